"""GPU checks at BASELINE.json's full sizes (configs 2-5), with data generated directly in HBM.

Where the oracle can still finish in seconds (navvis, 56.2 Mpoints; one doc file) the comparison is
exact; for the 2.6-Gpoint ca13 dataset the checks are size-independent properties: exact integer
partitions of a box, agreement between independent kernels (checksum of checksums), class-histogram
sums, chunking invariance of the grid (first-seen-wins across chunk boundaries) and an exact
oracle re-fold of a random sample of grid cells.
"""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
binding = importlib.import_module("adhoc-queries-pointclouds_amd.binding")
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")


class Resident:
    """Synthetic LAST column blocks of one file, generated in HBM."""

    def __init__(self, ctx, spec, want_cls=False):
        self.ctx, self.spec, self.n = ctx, spec, int(spec.n)
        self.xyz = ctx.alloc(12 * self.n)
        self.cls = ctx.alloc(self.n) if want_cls else None
        ctx.synth_fill(spec, 0, self.n, self.xyz, self.cls)
        self.h = specs.header_fields(spec)

    def cols(self, first=0, count=None):
        count = self.n - first if count is None else count
        return binding.make_columns(xyz=self.xyz + 12 * first, cls=(self.cls + first) if self.cls else None, n=count,
                                    first_index=first, scale=self.h["scale"], offset=self.h["offset"])

    def free(self):
        self.ctx.free(self.xyz)
        if self.cls:
            self.ctx.free(self.cls)


def count_bounds(ctx, r, lmin, lmax):
    cc = ctx.count_collector()
    ctx.scan_dev(r.cols(), pkg.Predicate.bounds(lmin, lmax), cc)
    n = cc.point_count()
    cc.free()
    return n


def test_config2_navvis_full_size_exact_vs_oracle(oracle, gpu_ctx):
    """BASELINE config 2: navvis (56.2 Mpoints, 1 file) --bounds S, count — exact against the oracle."""
    spec = specs.synth_navvis()[0]
    r = Resident(gpu_ctx, spec)
    try:
        image = oracle.synth_image(spec, transposed=True, threads=32)
        for q in ("navvis_S", "navvis_L", "navvis_XL"):
            bmin, bmax = specs.box(q)
            oc = oracle.count_collector()
            assert oracle.search_last_bounds(image, bmin, bmax, oc) == 0
            lmin, lmax = pkg.box_to_local(bmin, bmax, r.h["scale"], r.h["offset"])
            assert count_bounds(gpu_ctx, r, lmin, lmax) == oc.point_count(), q
            oc.free()
    finally:
        r.free()


def test_config3_doc_class_full_size(oracle, gpu_ctx):
    """BASELINE config 3: doc (8 x 106.75 Mpoints) --class 6.  One file exact vs the oracle; all files:
    the class histogram sums to N and class 19 never occurs (run_query_experiments.rs:332-343)."""
    ss = specs.synth_doc()
    total6 = total = 0
    for i, spec in enumerate(ss):
        r = Resident(gpu_ctx, spec, want_cls=True)
        try:
            hist = {}
            for c in (1, 2, 5, 6, 7, 9, 19, 0):
                cc = gpu_ctx.count_collector()
                gpu_ctx.scan_dev(r.cols(), pkg.Predicate.classification(c), cc)
                hist[c] = cc.point_count()
                cc.free()
            assert sum(hist.values()) == r.n and hist[19] == 0 and hist[0] == 0
            assert abs(hist[6] / r.n - 0.08) < 0.001
            total6 += hist[6]
            total += r.n
            if i == 0:
                _, cls = oracle.synth_columns(spec)  # the host generator's class column
                assert int((cls == 6).sum()) == hist[6]
                del cls
        finally:
            r.free()
    assert total == 854_000_000 and abs(total6 / total - 0.08) < 0.0005


@pytest.fixture(scope="module")
def ca13(gpu_ctx):
    files = [Resident(gpu_ctx, s) for s in specs.synth_ca13()]  # 16 x 163 M points, 31.3 GB of positions
    yield files
    for f in files:
        f.free()


def test_config5_ca13_full_size_properties(gpu_ctx, ca13):
    """ca13 (2 608 Mpoints): XL matches everything (run_query_experiments.rs:140); an integer box splits
    exactly; the batched launch and the per-file launches agree."""
    bmin, bmax = specs.box("ca13_XL")
    cols, preds, total_n = [], [], 0
    per_file = []
    for r in ca13:
        lmin, lmax = pkg.box_to_local(bmin, bmax, r.h["scale"], r.h["offset"])
        n = count_bounds(gpu_ctx, r, lmin, lmax)
        assert n == r.n  # every generated point lies inside the XL box
        per_file.append(n)
        cols.append(r.cols())
        preds.append(pkg.Predicate.bounds(lmin, lmax))
        total_n += r.n
    assert total_n == 2_608_000_000
    tot = gpu_ctx.alloc(16)
    gpu_ctx.memset(tot, 0, 16)
    gpu_ctx.scan_dev_count_batch(cols, preds, tot)
    host = np.zeros(1, dtype=np.uint64)
    gpu_ctx.to_host(host, tot)
    gpu_ctx.free(tot)
    assert int(host[0]) == sum(per_file) == total_n
    # exact integer partition of the L box of one file along every axis, through the per-file kernel
    r = ca13[5]
    bl, bh = specs.box("ca13_L")
    lmin, lmax = pkg.box_to_local(bl, bh, r.h["scale"], r.h["offset"])
    whole = count_bounds(gpu_ctx, r, lmin, lmax)
    assert 0 < whole < r.n
    for axis in range(3):
        lo, hi = max(lmin[axis], -2 ** 31), min(lmax[axis], 2 ** 31 - 1)
        mid = (lo + hi) // 2
        a_max, b_min = list(lmax), list(lmin)
        a_max[axis], b_min[axis] = mid, mid + 1
        assert count_bounds(gpu_ctx, r, lmin, a_max) + count_bounds(gpu_ctx, r, b_min, lmax) == whole


def test_config4_ca13_density_full_size(oracle, gpu_ctx, ca13):
    """BASELINE config 4: ca13 --bounds XL --density 10 on one file at full size (163 Mpoints, per-file grid).
    Properties: (1) scanning the file in one piece, in two halves and in 8 Mi-point chunks into one collector
    gives the identical winner set (first seen wins across chunk boundaries); (2) for a random sample of
    cells, re-folding ALL points of those cells with the oracle's SparseGrid gives the same winners."""
    r = ca13[3]
    if r.cls is None:  # result records carry the class byte (last.rs:138-142)
        r.cls = gpu_ctx.alloc(r.n)
        gpu_ctx.synth_fill(r.spec, 0, r.n, None, r.cls)
    bmin, bmax = specs.box("ca13_XL")
    lmin, lmax = pkg.box_to_local(bmin, bmax, r.h["scale"], r.h["offset"])
    pred = pkg.Predicate.bounds(lmin, lmax)

    def run(pieces):
        g = gpu_ctx.grid_collector(bmin, bmax, 10.0)
        for first, count in pieces:
            gpu_ctx.scan_dev(r.cols(first, count), pred, g)
        keys, pts = g.grid_cells(), g.points()
        dims, bits = g.grid_params()
        g.free()
        order = np.argsort(keys, kind="stable")
        return keys[order], pts[order], dims, bits

    n = r.n
    k1, p1, dims, bits = run([(0, n)])
    assert bits == [14, 14, 14] and dims == [9348, 9348, 9348]
    assert len(k1) == len(np.unique(k1)) and 0 < len(k1) <= n
    half = (n // 2) & ~3
    k2, p2, _, _ = run([(0, half), (half, n - half)])
    assert np.array_equal(k1, k2) and p1.tobytes() == p2.tobytes()
    step = 8 << 20
    k3, p3, _, _ = run([(f, min(step, n - f)) for f in range(0, n, step)])
    assert np.array_equal(k1, k3) and p1.tobytes() == p3.tobytes()

    # (2) exact re-fold of a sample of cells: numpy evaluates the same IEEE expressions for the keys
    xyz = np.empty((n, 3), dtype=np.int32)
    gpu_ctx.to_host(xyz, r.xyz)
    rng = np.random.default_rng(123)
    sample = np.sort(rng.choice(k1, size=64, replace=False))
    mn, mx = np.array(bmin), np.array(bmax)
    keys_all = np.zeros(n, dtype=np.uint64)
    shift = 0
    for a in range(3):
        w = xyz[:, a].astype(np.float64) * r.h["scale"][a] + r.h["offset"][a]  # (i*scale)+offset, unfused
        cell = (((w - mn[a]) * float(dims[a])) / (mx[a] - mn[a]))
        cell = np.where(cell > 0, cell, 0.0).astype(np.uint64)
        keys_all |= (cell & np.uint64((1 << bits[a]) - 1)) << np.uint64(shift)
        shift += bits[a]
    sel = np.nonzero(np.isin(keys_all, sample))[0]  # file order
    og = oracle.grid_collector(bmin, bmax, 10.0)
    for i in sel:
        og.collect_one(float(xyz[i, 0]) * r.h["scale"][0] + r.h["offset"][0], float(xyz[i, 1]) * r.h["scale"][1] + r.h["offset"][1],
                       float(xyz[i, 2]) * r.h["scale"][2] + r.h["offset"][2])
    assert np.array_equal(og.grid_cells(), sample)
    want = og.points()
    got = p1[np.isin(k1, sample)]
    for f in ("x", "y", "z"):
        assert np.array_equal(got[f], want[f])
    og.free()


def test_single_block_beyond_2_pow_32_points(gpu_ctx):
    """Index arithmetic at 64 bits: ONE positions block of 4.4 G points (52.8 GB; the largest LAS 1.2 point count
    is 2^32 - 1, LAS 1.4 goes beyond) through every count kernel.  Sized for the 288 GB of one MI355X."""
    n = (1 << 32) + 100_000_123
    spec = specs.synth_ca13(points_per_file=n, files=16)[5]
    r = Resident(gpu_ctx, spec, want_cls=True)
    try:
        h = r.h
        bmin, bmax = specs.box("ca13_XL")
        lmin, lmax = pkg.box_to_local(bmin, bmax, h["scale"], h["offset"])
        assert count_bounds(gpu_ctx, r, lmin, lmax) == n                       # the default one-wave kernel
        # a thin slab: x partition of the file's own extent (three disjoint pieces sum to n), per kernel family
        x0, x1 = lmin[0], lmax[0]
        lo, hi = int(h["min"][0] / h["scale"][0]), int(h["max"][0] / h["scale"][0])
        cut1, cut2 = lo + (hi - lo) // 3, lo + 2 * (hi - lo) // 3
        parts = [(x0, cut1), (cut1 + 1, cut2), (cut2 + 1, x1)]
        got = [count_bounds(gpu_ctx, r, [a, lmin[1], lmin[2]], [b, lmax[1], lmax[2]]) for a, b in parts]
        assert sum(got) == n and all(g > 0 for g in got)
        total = gpu_ctx.alloc(16)
        for bv in (3,):
            gpu_ctx.memset(total, 0, 16)
            gpu_ctx.scan_dev_count_batch([r.cols()] * 3, [pkg.Predicate.bounds([a, lmin[1], lmin[2]], [b, lmax[1], lmax[2]]) for a, b in parts], total)
            host = np.zeros(1, dtype=np.uint64)
            gpu_ctx.to_host(host, total)
            assert int(host[0]) == n, bv
        # the strided (generic) kernel: the same block addressed 4 bytes later with the last point dropped
        shifted = binding.make_columns(xyz=r.xyz + 12, n=n - 1, scale=h["scale"], offset=h["offset"])
        cc = gpu_ctx.count_collector()
        gpu_ctx.scan_dev(shifted, pkg.Predicate.bounds(lmin, lmax), cc)
        assert cc.point_count() == n - 1
        cc.free()
        # class histogram sums to n, per-file and batched kernels agree
        per_class = {}
        for c in (1, 2, 5, 6, 7, 9, 19):
            cc = gpu_ctx.count_collector()
            gpu_ctx.scan_dev(r.cols(), pkg.Predicate.classification(c), cc)
            per_class[c] = cc.point_count()
            cc.free()
        assert sum(per_class.values()) == n and per_class[19] == 0
        gpu_ctx.memset(total, 0, 16)
        gpu_ctx.scan_dev_count_batch([r.cols()] * 2, [pkg.Predicate.classification(6), pkg.Predicate.classification(2)], total)
        host = np.zeros(1, dtype=np.uint64)
        gpu_ctx.to_host(host, total)
        assert int(host[0]) == per_class[6] + per_class[2]
        gpu_ctx.free(total)
        # collectors that carry records: a very thin slab in the LAST quarter of the block (indices above 2^32)
        tail_first = n - 40_000_000
        tail = r.cols(first=tail_first)
        thin = pkg.Predicate.bounds([cut1, lmin[1], lmin[2]], [cut1 + 300, lmax[1], lmax[2]])
        cc, bc = gpu_ctx.count_collector(), gpu_ctx.buffer_collector()
        gpu_ctx.scan_dev(tail, thin, cc)
        gpu_ctx.scan_dev(tail, thin, bc)
        pts = bc.points()
        assert len(pts) == cc.point_count() > 0
        xs = np.rint(pts["x"] / h["scale"][0]).astype(np.int64)
        assert xs.min() >= cut1 and xs.max() <= cut1 + 300
        gbox = ([cut1 * h["scale"][0], h["min"][1], h["min"][2]], [(cut1 + 300) * h["scale"][0], h["max"][1], h["max"][2]])
        gc = gpu_ctx.grid_collector(gbox[0], gbox[1], 50.0)
        gpu_ctx.scan_dev(tail, thin, gc)
        cells = gc.point_count()
        gp = gc.points()
        assert 0 < cells <= len(pts) and len(gp) == cells
        assert set(map(bytes, gp.view(np.uint8).reshape(-1, 31))) <= set(map(bytes, pts.view(np.uint8).reshape(-1, 31)))  # winners are matches
        cc.free(), bc.free(), gc.free()
    finally:
        r.free()
