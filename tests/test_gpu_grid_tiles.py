"""GPU parity for the grid collector (csrc/grid_*.hip): pass 0 writes one block of tuples per tile of 5120 points and
folds a tile's duplicate cells before they travel; the fold reads the blocks back as per-bin fragment lists; tuples are
16 bytes (class byte — and, where two more top bytes are free, the second level's selector — inside the coordinates) or
24 (a colour column, a box wider than 2^24 units on every axis, a world-space predicate).  Same cells and winners as the oracle (grid_sampling.rs:49-105) in
every mode of the tile fold, on scan-ordered files with equal-distance ties that straddle tile boundaries, with aliased
keys inside tiles, and with runs of both tuple widths in one fold."""
import importlib

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
from test_gpu_scan import DevFile, small_spec  # noqa: E402

TILE = 5120  # P0_TILE of csrc/grid_common.h


def scan_ordered_image(oracle, seed, n, fmt, snap_step=512, snap_fraction=0.5):
    """A LAST image whose points are sorted along x inside y strips (scan lines, boustrophedon), half of them snapped to a
    coarse lattice — runs of identical positions with different attributes: equal distances, the first in file order must
    win — and with the last point of every tile repeated as the first point of the next one (a tie across the boundary)."""
    seed += int(os.environ.get("PCQ_TEST_SEED_BASE", "0"))  # a soak run: other seeds than the committed ones
    spec = small_spec(seed, n, fmt=fmt)
    image = oracle.synth_image(spec, transposed=True).copy()
    hdr = oracle.parse_header(image[:400].tobytes())
    otp = hdr.offset_to_point_data
    xyz = image[otp:otp + 12 * n].view("<i4").reshape(n, 3).copy()
    rng = np.random.default_rng(seed)
    snap = rng.random(n) < snap_fraction
    xyz[snap] = xyz[snap] // snap_step * snap_step
    strip = xyz[:, 1] // 400
    order = np.lexsort((np.where(strip % 2 == 0, xyz[:, 0], -xyz[:, 0]), strip))
    xyz = xyz[order]
    for k in range(TILE, n, TILE):
        xyz[k] = xyz[k - 1]
    image[otp:otp + 12 * n] = np.frombuffer(np.ascontiguousarray(xyz).tobytes(), dtype=np.uint8)
    return image, hdr


def check_same(gg, og):
    assert gg.point_count() == og.point_count()
    gp, gk = gg.points(), gg.grid_cells()
    order = np.argsort(gk, kind="stable")
    assert np.array_equal(gk[order], og.grid_cells())
    assert gp[order].tobytes() == og.points().tobytes()  # per cell: the same winner, every byte


@pytest.mark.parametrize("fmt", [1, 2])
@pytest.mark.parametrize("agg", [0, 1, 2])
def test_grid_tile_fold_on_scan_ordered_file_with_ties_across_tiles(oracle, agg, fmt):
    n = 40 * TILE + 1234
    image, hdr = scan_ordered_image(oracle, 20260 + fmt, n, fmt)
    with pkg.Context(0) as ctx:
        ctx.set_option("grid_agg", agg)
        f = DevFile(ctx, image, hdr)
        try:
            # a coarse grid (hundreds of consecutive points per cell), a finer one, one far smaller than the data with a
            # class query (cells beyond the key bits: aliased keys inside the tiles), and a dense one (second level)
            cases = [(20.0, (-60.0, -60.0, -12.0), (60.0, 60.0, 12.0), "bounds"),
                     (2.5, (-50.0, -50.0, -10.0), (50.0, 50.0, 10.0), "bounds"),
                     (4.0, (-16.0, -16.0, -4.0), (16.0, 16.0, 4.0), "class"),
                     (0.3, (-50.0, -50.0, -10.0), (50.0, 50.0, 10.0), "bounds")]
            for cell, bmin, bmax, kind in cases:
                og = oracle.grid_collector(bmin, bmax, cell)
                gg = ctx.grid_collector(bmin, bmax, cell)
                if kind == "bounds":
                    assert oracle.search_last_bounds(image, bmin, bmax, og) == 0
                    lmin, lmax = pkg.box_to_local(bmin, bmax, list(hdr.scale), list(hdr.offset))
                    ctx.scan_dev(f.columns(True), pkg.Predicate.bounds(lmin, lmax), gg)
                    matched = None
                else:
                    assert oracle.search_last_class(image, 2, og) == 0
                    ctx.scan_dev(f.columns(True), pkg.Predicate.classification(2), gg)
                    matched = None
                check_same(gg, og)
                tuples = ctx.get_option("grid_last_tuples")
                if cell == 20.0:
                    # 200 k matches into a few hundred cells, in scan order: the tile fold sheds most of them; without it
                    # every match travels
                    cc = ctx.count_collector()
                    ctx.scan_dev(f.columns(True), pkg.Predicate.bounds(lmin, lmax), cc)
                    matched = cc.point_count()
                    cc.free()
                    if agg == 2:
                        assert tuples == matched
                    else:
                        assert tuples < matched // 4, (tuples, matched)
                gg.free()
                og.free()
        finally:
            f.free()


@pytest.mark.parametrize("cell", [20.0, 0.25])
def test_grid_runs_of_both_tuple_widths_in_one_fold(oracle, cell):
    """Sequential mode (main.rs:129-133) shares one grid across files: a format-1 file (no colour: 20-byte tuples) and a
    format-2 file (24-byte tuples) meet in one fold — coarse: the big fold reads fragments of both widths; dense: the
    second level writes the wide form."""
    n = 700_001
    bmin, bmax = (-500.0, -500.0, -100.0), (500.0, 500.0, 100.0)
    images, hdrs = [], []
    for k, fmt in enumerate((1, 2, 1)):
        img = oracle.synth_image(small_spec(5150 + k, n, fmt=fmt), transposed=True)
        images.append(img)
        hdrs.append(oracle.parse_header(img[:400].tobytes()))
    og = oracle.grid_collector(bmin, bmax, cell)
    for img in images:
        assert oracle.search_last_bounds(img, bmin, bmax, og) == 0
    with pkg.Context(0) as ctx:
        gg = ctx.grid_collector(bmin, bmax, cell)
        files = []
        try:
            first = 0
            for img, hdr in zip(images, hdrs):
                f = DevFile(ctx, img, hdr)
                files.append(f)
                lmin, lmax = pkg.box_to_local(bmin, bmax, list(hdr.scale), list(hdr.offset))
                cols = f.columns(True)
                cols.first_index = first
                ctx.scan_dev(cols, pkg.Predicate.bounds(lmin, lmax), gg)
                first += n
            check_same(gg, og)
        finally:
            gg.free()
            for f in files:
                f.free()
    og.free()


@pytest.mark.parametrize("stream", [1, 0])
@pytest.mark.parametrize("tuple16", [1, 2, 0])
def test_grid_tuple_formats_and_fold_shapes_agree(oracle, tuple16, stream):
    """Round 4: 16-byte tuples with (1) and without (2) the second level's selector in their spare top bytes, or 24-byte
    tuples throughout (0); a coarse grid's bins folded as a stream (1) or by k_fold<BIG> (0).  Every combination gives the
    oracle's cells and winners — with ONE entry (the specialised kernels: no entry lookup, 16-byte loads), with TWO files of
    different scale in one grid (several entries: the tile -> entry table, the general kernels), with a class query (no class
    byte in the tuple), with a box wider than 2^24 units on every axis (no room for the class byte: 24 bytes), coarse and
    dense (second level: selector or recomputed cell)."""
    n = 300_007
    specs_ = [small_spec(8801, n, fmt=1, scale=(0.01, 0.01, 0.01)), small_spec(8802, n, fmt=1, scale=(0.02, 0.01, 0.005), offset=(1.0, -2.0, 0.5)),
              small_spec(8803, n, fmt=1, scale=(1e-5, 1e-5, 1e-5), lo=(-2_000_000_000, -2_000_000_000, -2_000_000_000), span=(4_000_000_000, 4_000_000_000, 4_000_000_000))]
    images = [oracle.synth_image(sp, transposed=True) for sp in specs_]
    hdrs = [oracle.parse_header(im[:400].tobytes()) for im in images]
    with pkg.Context(0) as ctx:
        ctx.set_option("grid_tuple16", tuple16)
        ctx.set_option("grid_stream", stream)
        files = [DevFile(ctx, im, h) for im, h in zip(images, hdrs)]
        try:
            cases = [("one entry, coarse", [0], 5.0, (-60.0, -60.0, -12.0), (60.0, 60.0, 12.0), "bounds", False),
                     ("one entry, dense", [0], 0.2, (-60.0, -60.0, -12.0), (60.0, 60.0, 12.0), "bounds", False),
                     ("two entries, coarse", [0, 1], 5.0, (-60.0, -60.0, -12.0), (60.0, 60.0, 12.0), "bounds", False),
                     ("two entries, dense", [0, 1], 0.2, (-60.0, -60.0, -12.0), (60.0, 60.0, 12.0), "bounds", False),
                     ("class query, coarse", [0, 1], 4.0, (-100.0, -100.0, -20.0), (120.0, 100.0, 20.0), "class", False),
                     ("class query, dense", [0], 0.3, (-60.0, -60.0, -12.0), (60.0, 60.0, 12.0), "class", False),
                     ("box wider than 2^24 units on every axis", [2], 2000.0, (-20000.0, -20000.0, -20000.0), (20000.0, 20000.0, 20000.0), "bounds", False),
                     # (round 4, late) 12 + 12 + 9 key bits: the hash's path for keys of more than 32 bits
                     ("keys wider than 32 bits", [0], 0.05, (-60.0, -60.0, -12.0), (60.0, 60.0, 12.0), "bounds", False),
                     # a fold between the scans: the second fold meets the first one's winners (they stay, or lose to a closer point)
                     ("the same file twice, coarse, folded in between", [0, 0], 5.0, (-60.0, -60.0, -12.0), (60.0, 60.0, 12.0), "bounds", True),
                     ("two entries, coarse, folded in between", [1, 0], 5.0, (-60.0, -60.0, -12.0), (60.0, 60.0, 12.0), "bounds", True),
                     ("two entries, dense, folded in between", [0, 1], 0.2, (-60.0, -60.0, -12.0), (60.0, 60.0, 12.0), "bounds", True)]
            for name, which, cell, bmin, bmax, kind, fold_between in cases:
                og = oracle.grid_collector(bmin, bmax, cell)
                gg = ctx.grid_collector(bmin, bmax, cell)
                first = 0
                for k in which:
                    cols = files[k].columns(True)
                    cols.first_index = first
                    first += n
                    if kind == "bounds":
                        assert oracle.search_last_bounds(images[k], bmin, bmax, og) == 0
                        lmin, lmax = pkg.box_to_local(bmin, bmax, list(hdrs[k].scale), list(hdrs[k].offset))
                        ctx.scan_dev(cols, pkg.Predicate.bounds(lmin, lmax), gg)
                    else:
                        assert oracle.search_last_class(images[k], 2, og) == 0
                        ctx.scan_dev(cols, pkg.Predicate.classification(2), gg)
                    if fold_between:
                        assert gg.point_count() == og.point_count(), name
                assert og.point_count() > 0, name
                check_same(gg, og)
                gg.free()
                og.free()
        finally:
            for f in files:
                f.free()


def test_grid_flush_folds_now_and_changes_nothing(oracle):
    """pcq_collector_flush: a per-file collector kept until all files are searched (main.rs:153-161) folds when its file
    is done; more scans may follow, the result is that of one fold at the end."""
    n = 300_007
    bmin, bmax = (-50.0, -50.0, -10.0), (50.0, 50.0, 10.0)
    cell = 1.5
    image = oracle.synth_image(small_spec(808, n, fmt=1), transposed=True)
    hdr = oracle.parse_header(image[:400].tobytes())
    og = oracle.grid_collector(bmin, bmax, cell)
    assert oracle.search_last_bounds(image, bmin, bmax, og) == 0
    assert oracle.search_last_bounds(image, bmin, bmax, og) == 0   # the same file again: every point loses its tie
    with pkg.Context(0) as ctx:
        f = DevFile(ctx, image, hdr)
        gg = ctx.grid_collector(bmin, bmax, cell)
        try:
            lmin, lmax = pkg.box_to_local(bmin, bmax, list(hdr.scale), list(hdr.offset))
            folds = ctx.get_option("grid_folds")
            cols = f.columns(True)
            ctx.scan_dev(cols, pkg.Predicate.bounds(lmin, lmax), gg)
            gg.flush()
            assert ctx.get_option("grid_folds") == folds + 1
            gg.flush()                                      # nothing pending: no fold
            assert ctx.get_option("grid_folds") == folds + 1
            cols.first_index = n
            ctx.scan_dev(cols, pkg.Predicate.bounds(lmin, lmax), gg)
            check_same(gg, og)
            assert ctx.get_option("grid_folds") == folds + 2
        finally:
            gg.free()
            f.free()
    og.free()


@pytest.mark.parametrize("fmt", [1, 2])
def test_grid_short_fragments_are_copied_together_first(oracle, fmt):
    """More than 1024 tiles and fewer than two tuples per (tile, bin) fragment — a box that few points match: the readers that
    keep a window of the bin's fragment list (the second level; k_fold<BIG>, the streaming fold's fallback) get the bins copied
    together first (grid_compactions); the streaming fold of a coarse grid reads the sparse run as it is."""
    n = 1100 * TILE + 77
    spec = small_spec(6061 + fmt, n, fmt=fmt)
    image = oracle.synth_image(spec, transposed=True)
    hdr = oracle.parse_header(image[:400].tobytes())
    bmin, bmax = (-30.0, -30.0, -2.0), (30.0, 30.0, 2.0)   # about 7 % of the points
    lmin, lmax = pkg.box_to_local(bmin, bmax, list(hdr.scale), list(hdr.offset))
    with pkg.Context(0) as ctx:
        f = DevFile(ctx, image, hdr)
        try:
            # coarse: the streaming fold on the sparse run itself / the same through k_fold<BIG> on the copied bins; dense (forced): the second level
            for cell, f2, stream, copied in ((2.0, 0, 1, 0), (2.0, 0, 0, 1), (0.1, 7, 1, 1)):
                ctx.set_option("grid_f2", f2)
                ctx.set_option("grid_stream", stream)
                og = oracle.grid_collector(bmin, bmax, cell)
                assert oracle.search_last_bounds(image, bmin, bmax, og) == 0
                before = ctx.get_option("grid_compactions")
                gg = ctx.grid_collector(bmin, bmax, cell)
                ctx.scan_dev(f.columns(True), pkg.Predicate.bounds(lmin, lmax), gg)
                check_same(gg, og)
                assert ctx.get_option("grid_compactions") == before + copied
                assert ctx.get_option("grid_last_f2") == (f2 or 1)
                gg.free()
                og.free()
            ctx.set_option("grid_stream", 1)
        finally:
            f.free()
