"""Lab kernels (libpcq_lab.so, PCQ_LAB=1): the experimental shapes of the grid collector that are meant to be CORRECT
must give the oracle's cells and winners too.  Skipped in the normal run (the product library has no variants)."""
import importlib
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("PCQ_LAB") != "1", reason="lab library only (PCQ_LAB=1)")]

pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
from test_gpu_scan import DevFile, small_spec  # noqa: E402


@pytest.mark.parametrize("variant", [8, 32, 128, 256, 512, 1024, 16384])
@pytest.mark.parametrize("cell", [1.0, 0.2])
def test_grid_lab_variant_matches_oracle(oracle, variant, cell):
    n = 2_000_003
    spec = small_spec(777 + variant, n, fmt=2)
    image = oracle.synth_image(spec, transposed=True)
    hdr = oracle.parse_header(image[:400].tobytes())
    bmin, bmax = (-45.0, -45.0, -9.0), (45.0, 45.0, 9.0)
    lmin, lmax = pkg.box_to_local(bmin, bmax, list(hdr.scale), list(hdr.offset))
    og = oracle.grid_collector(bmin, bmax, cell)
    assert oracle.search_last_bounds(image, bmin, bmax, og) == 0
    with pkg.Context(0) as ctx:
        ctx.set_option("grid_variant", variant)
        f = DevFile(ctx, image, hdr)
        try:
            gg = ctx.grid_collector(bmin, bmax, cell)
            ctx.scan_dev(f.columns(True), pkg.Predicate.bounds(lmin, lmax), gg)
            assert gg.point_count() == og.point_count()
            gp, gk = gg.points(), gg.grid_cells()
            order = np.argsort(gk, kind="stable")
            assert np.array_equal(gk[order], og.grid_cells())
            assert gp[order].tobytes() == og.points().tobytes()
            gg.free()
        finally:
            f.free()
    og.free()
