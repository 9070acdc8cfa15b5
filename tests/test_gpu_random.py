"""Randomised GPU parity at the file level: LAS / LAST images of every point format (0-10, LAS 1.2 and
1.4 headers, extra bytes, VLR padding, odd record lengths, anisotropic scales, coordinate extremes,
deliberately wrong header bounds) are built with numpy — independently of the oracle's generator —
and pushed through the product (libpcq_query.so -> HIP) and the oracle with random queries.
Counts, ordered result records and per-cell grid winners must be identical; errors must agree.
"""
import ctypes as C
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from test_gpu_host import Q  # noqa: E402  (ctypes view of include/pcq_query.h)

SEED_BASE = int(os.environ.get("PCQ_TEST_SEED_BASE", "0"))  # a soak run: other seeds than the committed ones
FORMAT_LEN = {0: 20, 1: 28, 2: 26, 3: 34, 4: 57, 5: 63, 6: 30, 7: 36, 8: 38, 9: 59, 10: 67}
COLOR_AT = {2: 20, 3: 28, 5: 28}  # the colour offsets the optimized scans know (last.rs:83-88)


def build(rng, transposed):
    fmt = int(rng.integers(0, 11))
    v14 = fmt >= 6 or rng.random() < 0.2
    header_size = 375 if v14 else 227
    extra = int(rng.choice([0, 0, 1, 3, 7]))
    rl = FORMAT_LEN[fmt] + extra
    otp = header_size + int(rng.choice([0, 0, 54, 101]))
    n = int(rng.choice([0, 1, 2, 63, 64, 65, 255, 256, 257, 1000, 2047, 2048, 2049, int(rng.integers(3000, 20000))]))
    scale = rng.choice([0.001, 0.01, 0.1, 0.25, 0.5, 1.0], size=3)
    offset = rng.choice([0.0, 100.0, -2500.5, 389400.0, 1e6], size=3)
    span = rng.choice([50, 2000, 100000, 2 ** 31 - 1], size=3)
    centre = rng.integers(-1000, 1000, size=3)
    lo = np.maximum(centre - span // 2, -2 ** 31)
    hi = np.minimum(centre + span // 2, 2 ** 31 - 1)
    xyz = np.stack([rng.integers(lo[a], hi[a] + 1, size=n) for a in range(3)], axis=1).astype("<i4")
    if n >= 4:  # extremes and duplicates
        xyz[0] = [2 ** 31 - 1, -2 ** 31, 0]
        xyz[1] = xyz[2]
    cls = rng.choice(np.array([0, 1, 2, 6, 6, 134, 255], dtype=np.uint8), size=n)
    rgb = rng.integers(0, 65536, size=(n, 3)).astype("<u2")
    cls_at = 15 if fmt <= 5 else 16

    img = bytearray(otp + n * rl)
    img[0:4] = b"LASF"
    img[24], img[25] = 1, (4 if v14 else 2)
    struct.pack_into("<H", img, 94, header_size)
    struct.pack_into("<I", img, 96, otp)
    img[104] = fmt
    struct.pack_into("<H", img, 105, rl)
    legacy = n if (not v14 or rng.random() < 0.5) else 0
    struct.pack_into("<I", img, 107, legacy)
    struct.pack_into("<ddd", img, 131, *scale)
    struct.pack_into("<ddd", img, 155, *offset)
    world = xyz.astype(np.float64) * scale + offset
    if n and rng.random() < 0.8:
        wmin, wmax = world.min(axis=0), world.max(axis=0)
    else:  # a header box that does not describe the data: the early-out must follow the HEADER (last.rs:92-94)
        wmin, wmax = np.array([0.0, 0.0, 0.0]), np.array([10.0, 10.0, 10.0])
    for a in range(3):
        struct.pack_into("<dd", img, 179 + 16 * a, wmax[a], wmin[a])
    if v14:
        struct.pack_into("<Q", img, 247, n)
    buf = np.frombuffer(img, dtype=np.uint8).copy()
    if n:
        body = buf[otp:]
        if transposed:  # LAST: attribute blocks at otp + N * offset_in_record (last_reader.rs:83-144)
            body[:12 * n] = xyz.view(np.uint8).reshape(-1)
            body[cls_at * n:(cls_at + 1) * n] = cls
            if fmt in COLOR_AT:
                body[COLOR_AT[fmt] * n:COLOR_AT[fmt] * n + 6 * n] = rgb.view(np.uint8).reshape(-1)
        else:
            rec = body.reshape(n, rl)
            rec[:, :12] = xyz.view(np.uint8).reshape(n, 12)
            rec[:, 15] = rng.integers(0, 256, size=n)  # the LAS bounds path ALWAYS reads +15 (las.rs:121-124)
            rec[:, cls_at] = cls
            if fmt in COLOR_AT:
                rec[:, COLOR_AT[fmt]:COLOR_AT[fmt] + 6] = rgb.view(np.uint8).reshape(n, 6)
    return buf, world, dict(fmt=fmt, n=n, rl=rl, otp=otp, v14=v14)


def random_box(rng, world):
    if len(world) and rng.random() < 0.85:
        c = world[rng.integers(0, len(world))]
        half = np.abs(rng.choice([0.0, 0.004, 0.3, 7.0, 150.0, 1e5, 1e10], size=3))
        lo, hi = c - half, c + half * rng.choice([0.0, 1.0, 1.0], size=3)
    else:
        lo = rng.uniform(-1e4, 1e4, size=3)
        hi = lo + rng.uniform(0, 1e4, size=3)
    return [float(v) for v in lo], [float(v) for v in np.maximum(lo, hi)]


def sorted_grid(q, h):
    keys, pts = q.cells(h), q.points(h)
    order = np.argsort(keys, kind="stable")
    return keys[order], pts[order]


@pytest.mark.parametrize("seed", range(150))
def test_random_files_and_queries(oracle, tmp_path, seed):
    rng = np.random.default_rng(1000 + SEED_BASE + seed)
    q = Q()
    transposed = bool(seed % 2)
    image, world, meta = build(rng, transposed)
    path = str(tmp_path / ("f.last" if transposed else "f.las"))
    image.tofile(path)
    for _ in range(4):
        bmin, bmax = random_box(rng, world)
        # count + buffer
        oc, ob = oracle.count_collector(), oracle.buffer_collector()
        rc_o, rec_o = oracle.search_file(path, 0, bmin, bmax, 0, oc)
        oracle.search_file(path, 0, bmin, bmax, 0, ob)
        hc, hb = q.collector("count"), q.collector("buffer")
        rc_c, rec_c = q.search_bounds(path, bmin, bmax, hc)
        rc_b, _ = q.search_bounds(path, bmin, bmax, hb)
        assert rc_c == rc_o == rc_b, (meta, bmin, bmax, q.lib.pcq_query_last_error())
        assert rec_c == rec_o
        if rc_o == 0:
            assert q.count(hc) == oc.point_count(), (meta, bmin, bmax)
            assert q.points(hb).tobytes() == ob.points().tobytes(), (meta, bmin, bmax)
        q.free(hc), q.free(hb), oc.free(), ob.free()
        # grid over the same box
        cell = float(rng.choice([0.05, 1.0, 12.5, 1000.0, 1e9]))
        try:
            og = oracle.grid_collector(bmin, bmax, cell)
        except Exception:
            og = None
        h = C.c_void_p()
        rc_new = q.lib.pcq_query_collector_new_grid(0, q.d3(bmin), q.d3(bmax), cell, C.byref(h))
        if og is None:
            assert rc_new == -6  # SparseGrid::new: too many cells
            continue
        if rc_new == -11:  # 64 key bits / non-finite: documented unsupported corner
            og.free()
            continue
        assert rc_new == 0
        assert oracle.search_file(path, 0, bmin, bmax, 0, og)[0] == q.search_bounds(path, bmin, bmax, h)[0]
        gk, gp = sorted_grid(q, h)
        assert np.array_equal(gk, og.grid_cells()), (meta, bmin, bmax, cell)
        assert gp.tobytes() == og.points().tobytes(), (meta, bmin, bmax, cell)
        q.free(h), og.free()
    for cls in (6, 134, 0, 19):
        oc, ob = oracle.count_collector(), oracle.buffer_collector()
        rc_o = oracle.search_file(path, 1, None, None, cls, oc)[0]
        oracle.search_file(path, 1, None, None, cls, ob)
        hc, hb = q.collector("count"), q.collector("buffer")
        assert q.search_class(path, cls, hc) == rc_o and q.search_class(path, cls, hb) == rc_o, meta
        if rc_o == 0:
            assert q.count(hc) == oc.point_count(), (meta, cls)
            assert q.points(hb).tobytes() == ob.points().tobytes(), (meta, cls)
        q.free(hc), q.free(hb), oc.free(), ob.free()


@pytest.mark.parametrize("seed", range(60))
def test_random_lazer_files_and_queries(oracle, tmp_path, seed):
    """The same randomly built point sets as LAZER files (random block size, LZ4 frame flags, every point
    format): world-space `contains` on the GPU vs the oracle's streaming restatement, boxes whose faces
    pass exactly through points included; n = 0 panics on both sides (lazer_reader.rs:123,143)."""
    rng = np.random.default_rng(7000 + SEED_BASE + seed)
    q = Q()
    image, world, meta = build(rng, True)
    if meta["n"] == 0:
        blocks = [1]
    else:
        blocks = [int(rng.choice([1, 2, 63, 64, 100, 1000, 4096, max(1, meta["n"] // 3), meta["n"], meta["n"] + 5]))]
    if meta["n"] > 3000:
        blocks = [b for b in blocks if b >= 16] or [1000]
    flags = int(rng.choice([0, 1, 2, 4, 8, 1 | 2 | 4 | 8, 16, 16 | 2, 4 | 8]))
    lazer = oracle.lazer_from_last(image, blocks[0], flags, int(rng.choice([4, 5])))
    path = str(tmp_path / "f.lazer")
    lazer.tofile(path)
    for _ in range(4):
        bmin, bmax = random_box(rng, world)
        oc, ob = oracle.count_collector(), oracle.buffer_collector()
        rc_o = oracle.search_file(path, 0, bmin, bmax, 0, oc)[0]
        oracle.search_file(path, 0, bmin, bmax, 0, ob)
        hc, hb = q.collector("count"), q.collector("buffer")
        rc_c = q.search_bounds(path, bmin, bmax, hc)[0]
        rc_b = q.search_bounds(path, bmin, bmax, hb)[0]
        assert rc_c == rc_o == rc_b, (meta, blocks, flags, q.lib.pcq_query_last_error())
        if rc_o == 0:
            assert q.count(hc) == oc.point_count(), (meta, bmin, bmax)
            assert q.points(hb).tobytes() == ob.points().tobytes(), (meta, bmin, bmax)
        q.free(hc), q.free(hb), oc.free(), ob.free()
        cell = float(rng.choice([0.05, 1.0, 12.5, 1000.0]))
        try:
            og = oracle.grid_collector(bmin, bmax, cell)
        except Exception:
            continue
        h = C.c_void_p()
        rc_new = q.lib.pcq_query_collector_new_grid(0, q.d3(bmin), q.d3(bmax), cell, C.byref(h))
        if rc_new != 0:
            og.free()
            continue
        assert oracle.search_file(path, 0, bmin, bmax, 0, og)[0] == q.search_bounds(path, bmin, bmax, h)[0]
        gk, gp = sorted_grid(q, h)
        assert np.array_equal(gk, og.grid_cells()), (meta, bmin, bmax, cell)
        assert gp.tobytes() == og.points().tobytes(), (meta, bmin, bmax, cell)
        q.free(h), og.free()
    for cls in (6, 134, 19):
        oc, ob = oracle.count_collector(), oracle.buffer_collector()
        rc_o = oracle.search_file(path, 1, None, None, cls, oc)[0]
        oracle.search_file(path, 1, None, None, cls, ob)
        hc, hb = q.collector("count"), q.collector("buffer")
        assert q.search_class(path, cls, hc) == rc_o and q.search_class(path, cls, hb) == rc_o, (meta, blocks, flags)
        if rc_o == 0:
            assert q.count(hc) == oc.point_count(), (meta, cls)
            assert q.points(hb).tobytes() == ob.points().tobytes(), (meta, cls)
        q.free(hc), q.free(hb), oc.free(), ob.free()
