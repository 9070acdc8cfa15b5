#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/.

The reference (Rust) cannot be built or run in the container and its own tests pin only SparseGrid
(query/src/grid_sampling.rs:116-209), so the known-answer vectors here are derived by evaluating
the reference's expressions literally with Python's IEEE-754 doubles and arbitrary-precision
integers — a restatement independent of oracle/*.c — on inputs small enough to check by hand.
Each block cites the reference lines it evaluates.  Output: expected.json, tiny_fmt2.last,
tiny_fmt3.las (a few hundred bytes each).

Run:  python tests/golden/make_golden.py      (deterministic; rewrites the files in place)
"""
import json
import math
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))

I64_MIN, I64_MAX, U64_MAX = -(2 ** 63), 2 ** 63 - 1, 2 ** 64 - 1


def as_i64(v: float) -> int:  # Rust `f64 as i64` (>= 1.45): truncate, saturate, NaN -> 0
    if v != v:
        return 0
    if v >= 2.0 ** 63:
        return I64_MAX
    if v <= -(2.0 ** 63):
        return I64_MIN
    return int(v)


def as_u64(v: float) -> int:  # Rust `f64 as u64`
    if v != v or v <= 0.0:
        return 0
    if v >= 2.0 ** 64:
        return U64_MAX
    return int(v)


def box_to_local(bmin, bmax, scale, offset):  # last.rs:98-109 — x scale on all three min components
    lmin = [as_i64((bmin[a] - offset[a]) / scale[0]) for a in range(3)]
    lmax = [as_i64((bmax[a] - offset[a]) / scale[a]) for a in range(3)]
    panic = any(lmin[a] > lmax[a] for a in range(3))  # AABB::from_min_max [recalled]
    return lmin, lmax, panic


def intersects(amin, amax, bmin, bmax):  # pasture AABB::intersects (inclusive), last.rs:92
    return all(amin[a] <= bmax[a] and amax[a] >= bmin[a] for a in range(3))


def world(i, scale, offset):  # last.rs:156-160: (i as f64 * scale) + offset
    return (float(i) * scale) + offset


# ---- SparseGrid (grid_sampling.rs:17-105) ---------------------------------------------------------
class Grid:
    def __init__(self, bmin, bmax, cell):
        self.bmin, self.bmax, self.cell = bmin, bmax, cell
        self.dims, self.bits = [], []
        for a in range(3):
            ncells = math.ceil((bmax[a] - bmin[a]) / cell) if math.isfinite((bmax[a] - bmin[a]) / cell) else (bmax[a] - bmin[a]) / cell
            ncells = float(ncells)
            lg = math.log2(ncells) if ncells > 0 else (float("-inf") if ncells == 0 else float("nan"))
            self.bits.append(as_u64(float(math.ceil(lg)) if math.isfinite(lg) else lg))
            self.dims.append(as_u64(ncells))
        self.too_many = sum(self.bits) > 64
        self.cells = {}  # key -> point tuple (insertion keeps the reference's HashMap semantics)

    def insert(self, pt):
        cell = []
        for a in range(3):
            r = (pt[a] - self.bmin[a]) * float(self.dims[a]) / (self.bmax[a] - self.bmin[a])  # :51-56
            cell.append(as_u64(r))  # :58-60
        mask = [((1 << (b & 63)) - 1) for b in self.bits]  # :62-64 (release-mode shl masks the amount)
        ys, zs = self.bits[0] & 63, (self.bits[0] + self.bits[1]) & 63
        key = (cell[0] & mask[0]) | (((cell[1] & mask[1]) << ys) & U64_MAX) | (((cell[2] & mask[2]) << zs) & U64_MAX)
        if key not in self.cells:  # :73-76
            self.cells[key] = pt
            return key
        centre = [(float(cell[a]) + 0.5) * self.cell + self.bmin[a] for a in range(3)]  # :78-82 unmasked cell

        def d2(p):  # nalgebra distance_squared: dx*dx + dy*dy + dz*dz, (a+b)+c
            dx, dy, dz = p[0] - centre[0], p[1] - centre[1], p[2] - centre[2]
            return (dx * dx + dy * dy) + dz * dz

        if d2(pt) < d2(self.cells[key]):  # :97 strict
            self.cells[key] = pt
        return key


# ---- tiny files -------------------------------------------------------------------------------------
SCALE = (0.01, 0.02, 0.05)   # anisotropic on purpose: exposes the x_scale typo on the min corner
OFFSET = (100.0, 200.0, -10.0)
# (X, Y, Z, class, (R, G, B))
POINTS = [
    (0, 0, 0, 2, (1, 2, 3)),
    (1000, 500, 200, 6, (10, 20, 30)),           # world (110, 210, 0)
    (1001, 500, 200, 6, (11, 21, 31)),           # one x unit past the box max below
    (-1070, 0, 0, 2, (12, 22, 32)),
    (500, 250, 100, 6, (13, 23, 33)),            # world (105, 205, -5)
    (500, 250, 100, 134, (14, 24, 34)),          # same position, class byte 6 with a flag bit set (0x86)
    (2147483647, -2147483648, 0, 1, (15, 25, 35)),
    (999, 499, 199, 9, (16, 26, 36)),
    (0, 501, 0, 6, (17, 27, 37)),
    (250, 125, 50, 5, (65535, 0, 65535)),
    (750, 375, 150, 6, (18, 28, 38)),
    (1, 1, 1, 19, (19, 29, 39)),
]


def header(fmt, rec_len, n, pts_world):
    h = bytearray(227)
    h[0:4] = b"LASF"
    h[24], h[25] = 1, 2
    h[26:35] = b"pcq-golden"[:9]
    h[58:67] = b"pcq-golden"[:9]
    struct.pack_into("<HH", h, 90, 1, 2026)
    struct.pack_into("<H", h, 94, 227)
    struct.pack_into("<I", h, 96, 227)
    struct.pack_into("<I", h, 100, 0)
    h[104] = fmt
    struct.pack_into("<H", h, 105, rec_len)
    struct.pack_into("<I", h, 107, n)
    struct.pack_into("<I", h, 111, n)
    struct.pack_into("<ddd", h, 131, *SCALE)
    struct.pack_into("<ddd", h, 155, *OFFSET)
    for a in range(3):
        struct.pack_into("<dd", h, 179 + 16 * a, max(p[a] for p in pts_world), min(p[a] for p in pts_world))
    return bytes(h)


def build_files():
    n = len(POINTS)
    pts_world = [tuple(world(p[a], SCALE[a], OFFSET[a]) for a in range(3)) for p in POINTS]
    # LAST, format 2 (record 26 B): XYZ@0 (12) I@12 (2) bits@14 cls@15 angle@16 user@17 psid@18 (2) RGB@20 (6)
    last = bytearray(header(2, 26, n, pts_world)) + bytearray(26 * n)
    base = 227
    for i, (x, y, z, c, rgb) in enumerate(POINTS):
        struct.pack_into("<iii", last, base + 12 * i, x, y, z)
        struct.pack_into("<H", last, base + 12 * n + 2 * i, 100 + i)
        last[base + 14 * n + i] = 0x11
        last[base + 15 * n + i] = c
        struct.pack_into("<HHH", last, base + 20 * n + 6 * i, *rgb)
    # LAS, format 3 (record 34 B): ... gps@20 (8) RGB@28 (6)
    las = bytearray(header(3, 34, n, pts_world)) + bytearray(34 * n)
    for i, (x, y, z, c, rgb) in enumerate(POINTS):
        o = base + 34 * i
        struct.pack_into("<iii", las, o, x, y, z)
        struct.pack_into("<H", las, o + 12, 100 + i)
        las[o + 14] = 0x11
        las[o + 15] = c
        struct.pack_into("<d", las, o + 20, 0.5 * i)
        struct.pack_into("<HHH", las, o + 28, *rgb)
    return bytes(last), bytes(las), pts_world


def scan_bounds(bmin, bmax, pts_world, has_color=True):
    hmin = [min(p[a] for p in pts_world) for a in range(3)]
    hmax = [max(p[a] for p in pts_world) for a in range(3)]
    if not intersects(hmin, hmax, bmin, bmax):  # last.rs:92-94
        return {"skipped": True, "indices": [], "panic": False}
    lmin, lmax, panic = box_to_local(bmin, bmax, SCALE, OFFSET)
    if panic:
        return {"skipped": False, "indices": [], "panic": True}
    idx = [i for i, p in enumerate(POINTS) if all(lmin[a] <= p[a] <= lmax[a] for a in range(3))]  # :122-135
    return {"skipped": False, "indices": idx, "panic": False, "lmin": lmin, "lmax": lmax}


def records(indices, pts_world):
    return [[pts_world[i][0].hex(), pts_world[i][1].hex(), pts_world[i][2].hex(), *POINTS[i][4], POINTS[i][3]] for i in indices]


def main():
    out = {}
    # ---- casts --------------------------------------------------------------------------------------
    cast_inputs = [0.0, -0.0, 0.7, -0.7, 10.7, -10.7, 1e19, -1e19, 2.0 ** 63, -(2.0 ** 63), 2.0 ** 64, 1.8446744073709552e19,
                   float("nan"), float("inf"), float("-inf"), 9007199254740993.0, 4294967295.9, -1.0, 1.0e-320]
    out["casts"] = [{"f": v.hex() if v == v else "nan", "i64": as_i64(v), "u64": as_u64(v)} for v in cast_inputs]

    # ---- box conversion (last.rs:98-109) ---------------------------------------------------------------
    box_cases = [
        ((100.0, 200.0, -10.0), (110.0, 210.0, 0.0), SCALE, OFFSET),                 # exact edges
        ((99.893, 199.3, -10.7), (110.007, 210.7, 0.7), SCALE, OFFSET),              # truncation toward zero of +-x.7
        ((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (0.001, 0.01, 0.1), (0.0, 0.0, 0.0)),     # typo: min y,z divided by x scale
        ((-5.5, -5.5, -5.5), (5.5, 5.5, 5.5), (0.5, 0.25, 2.0), (1.0, -1.0, 0.0)),
        ((-1e30, -1e30, -1e30), (1e30, 1e30, 1e30), (0.01, 0.01, 0.01), (0.0, 0.0, 0.0)),   # saturation
        ((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (0.001, 0.01, 0.01), (0.0, 0.0, 0.0)),    # min.y = 0 ok; just anisotropy
        ((5.0, 5.0, 5.0), (6.0, 6.0, 6.0), (0.001, 0.1, 0.1), (0.0, 0.0, 0.0)),      # typo makes lmin.y > lmax.y -> panic
        ((643431.76, 3883547.565, -46194.145), (736910.93, 3977026.735, 47285.025), (0.01, 0.01, 0.01), (0.0, 0.0, 0.0)),  # ca13 XL
        ((float("nan"), 0.0, 0.0), (1.0, 1.0, 1.0), (0.01, 0.01, 0.01), (0.0, 0.0, 0.0)),   # NaN -> 0
    ]
    out["box_to_local"] = []
    for bmin, bmax, sc, off in box_cases:
        lmin, lmax, panic = box_to_local(bmin, bmax, sc, off)
        out["box_to_local"].append({"bmin": [v.hex() if v == v else "nan" for v in bmin], "bmax": [v.hex() for v in bmax],
                                    "scale": [v.hex() for v in sc], "offset": [v.hex() for v in off], "lmin": lmin, "lmax": lmax,
                                    "panic": panic})

    # ---- tiny files ---------------------------------------------------------------------------------------
    last, las, pts_world = build_files()
    open(os.path.join(HERE, "tiny_fmt2.last"), "wb").write(last)
    open(os.path.join(HERE, "tiny_fmt3.las"), "wb").write(las)
    queries = {
        "box_exact": ((100.0, 200.0, -10.0), (110.0, 210.0, 0.0)),
        "box_everything": ((-1e9, -1e9, -1e9), (1e9, 1e9, 1e9)),
        "box_typo": ((100.0, 205.0, -10.0), (110.0, 210.0, 0.0)),    # min.y local = 5/0.01 = 500 (not 250): y >= 500 only
        "box_miss": ((500.0, 500.0, 500.0), (600.0, 600.0, 600.0)),  # header AABB early-out... but the int32 extreme point widens the header box
        "box_far_miss": ((3e7, 0.0, 0.0), (4e7, 1.0, 1.0)),
        "box_edge_touch": ((110.0, 210.0, 0.0), (120.0, 220.0, 10.0)),  # touches the (1000,500,200) point inclusively
        "box_negative_trunc": ((89.295, 199.99, -10.04), (100.004, 200.01, -9.96)),
    }
    out["tiny"] = {"scale": list(SCALE), "offset": list(OFFSET), "n": len(POINTS), "bounds": {}, "class": {}, "grid": {}}
    for name, (bmin, bmax) in queries.items():
        r = scan_bounds(bmin, bmax, pts_world)
        r["bmin"], r["bmax"] = list(bmin), list(bmax)
        r["records"] = records(r["indices"], pts_world)
        out["tiny"]["bounds"][name] = r
    for c in (6, 2, 134, 19, 0, 255):
        idx = [i for i, p in enumerate(POINTS) if p[3] == c]  # last.rs:259-262 whole byte
        out["tiny"]["class"][str(c)] = {"indices": idx, "records": records(idx, pts_world)}

    # grid over a bounds query: box_exact with cell 2.5 (dims 4x4x4, 2 bits per axis -> cell 4 aliases to 0)
    for gname, (qname, cell) in {"exact_2.5": ("box_exact", 2.5), "exact_3": ("box_exact", 3.0), "everything_1e8": ("box_everything", 1e8)}.items():
        bmin, bmax = queries[qname]
        g = Grid(list(bmin), list(bmax), cell)
        idxs = out["tiny"]["bounds"][qname]["indices"]
        owner = {}
        for i in idxs:
            pt = pts_world[i]
            k = g.insert((pt[0], pt[1], pt[2], i))
        cells = sorted(g.cells.items())
        out["tiny"]["grid"][gname] = {"query": qname, "cell": cell, "dims": g.dims, "bits": g.bits,
                                      "keys": [k for k, _ in cells], "winners": [v[3] for _, v in cells]}

    # ---- the reference's own SparseGrid tests (grid_sampling.rs:121-208) + extra traps -------------------------
    def run_grid(bmin, bmax, cell, pts):
        g = Grid(bmin, bmax, cell)
        for i, p in enumerate(pts):
            g.insert((p[0], p[1], p[2], i))
        cells = sorted(g.cells.items())
        return {"bmin": bmin, "bmax": bmax, "cell": cell, "points": [list(p) for p in pts], "dims": g.dims, "bits": g.bits,
                "too_many": g.too_many, "keys": [k for k, _ in cells], "winners": [v[3] for _, v in cells]}

    b5 = ([-5.0] * 3, [5.0] * 3)
    out["grid"] = {
        "ref_add_one": run_grid(*b5, 1.0, [(-4.5, -4.6, -4.7)]),                                   # :121-143 -> key 0
        "ref_different_cells": run_grid(*b5, 1.0, [(-4.5, -4.6, -4.7), (-3.5, -4.5, -4.4)]),        # :146-179 -> keys {0,1}
        "ref_same_cell": run_grid(*b5, 1.0, [(-4.8, -4.6, -4.7), (-4.5, -4.4, -4.6)]),              # :182-208 -> second wins
        "tie_first_wins": run_grid(*b5, 1.0, [(-4.25, -4.5, -4.5), (-4.75, -4.5, -4.5), (-4.5, -4.25, -4.5)]),
        "alias_pow2_dims": run_grid([0.0] * 3, [8.0] * 3, 1.0, [(0.9, 0.5, 0.5), (8.0, 0.5, 0.5), (0.5, 0.5, 0.5), (8.0, 0.4, 0.5)]),
        "alias_order_dependent": run_grid([0.0] * 3, [8.0] * 3, 1.0, [(8.0, 0.5, 0.5), (0.1, 0.5, 0.5), (8.4, 0.5, 0.5), (0.45, 0.5, 0.5)]),
        "negative_saturates_to_cell0": run_grid(*b5, 1.0, [(-7.0, -4.5, -4.5), (-4.6, -4.5, -4.5)]),
        "non_pow2_dims_no_alias": run_grid([0.0] * 3, [10.0] * 3, 1.0, [(10.0, 0.5, 0.5), (0.5, 0.5, 0.5)]),
        "dims_one_zero_bits": run_grid([0.0] * 3, [1.0, 4.0, 1.0], 1.0, [(0.2, 3.5, 0.2), (0.6, 0.5, 0.5), (0.5, 3.4, 0.5)]),
    }
    # the same traps with every coordinate on the 1/64 lattice, so that a LAS file with scale 1/64 rebuilds
    # the positions exactly and the vectors can also be pushed through the file-level (GPU) path
    L = 1.0 / 64.0
    out["grid"].update({
        "lat_ref_same_cell": run_grid(*b5, 1.0, [(-4.75, -4.625, -4.6875), (-4.5, -4.375, -4.59375)]),
        "lat_tie_first_wins": run_grid(*b5, 1.0, [(-4.25, -4.5, -4.5), (-4.75, -4.5, -4.5), (-4.5, -4.25, -4.5), (-4.5, -4.5, -4.75)]),
        "lat_alias_pow2_dims": run_grid([0.0] * 3, [8.0] * 3, 1.0, [(0.875, 0.5, 0.5), (8.0, 0.5, 0.5), (0.5, 0.5, 0.5), (8.0, 0.375, 0.5)]),
        "lat_alias_order_dependent": run_grid([0.0] * 3, [8.0] * 3, 1.0, [(8.0, 0.5, 0.5), (0.125, 0.5, 0.5), (8.375, 0.5, 0.5),
                                                                          (0.453125, 0.5, 0.5), (8.5, 0.5, 0.5), (0.5, 0.5, 0.5)]),
        "lat_alias_two_axes": run_grid([0.0] * 3, [4.0] * 3, 1.0, [(4.0, 4.0, 0.5), (0.25, 0.25, 0.5), (4.5, 0.5, 0.5), (0.5, 4.5, 0.5),
                                                                   (1.5, 0.5, 0.5), (0.5, 0.5, 0.5), (4.25, 4.25, 0.25)]),
        "lat_negative_saturates": run_grid(*b5, 1.0, [(-7.0, -4.5, -4.5), (-4.625, -4.5, -4.5), (-9.0, -9.0, -9.0)]),
        "lat_non_pow2_dims_no_alias": run_grid([0.0] * 3, [10.0] * 3, 1.0, [(10.0, 0.5, 0.5), (0.5, 0.5, 0.5), (10.0, 0.25, 0.5)]),
    })
    assert all(abs(c / L - round(c / L)) == 0 for k, v in out["grid"].items() if k.startswith("lat_") for p in v["points"] for c in p)
    g_big = Grid([0.0] * 3, [1e9] * 3, 1e-4)  # 3 x 44 bits > 64
    out["grid"]["too_many_cells"] = {"bmin": [0.0] * 3, "bmax": [1e9] * 3, "cell": 1e-4, "bits": g_big.bits, "too_many": g_big.too_many}

    lazer_section(out, last, pts_world)
    json.dump(out, open(os.path.join(HERE, "expected.json"), "w"), indent=1)
    print("wrote expected.json, tiny_fmt2.last (%d B), tiny_fmt3.las (%d B)" % (len(last), len(las)))


# ---- LAZER (readers/src/lazer_reader.rs, query/src/search/lazer.rs) ------------------------------------------
# The column blobs are compressed by the image's real liblz4 (tests/_lz4ref.py), the library under the
# reference's `lz4` crate — not by anything in oracle/ or the product.  Without liblz4 on the host the
# committed files are left as they are.
LAZER_BLOCK = 7


def contains(bmin, bmax, p):  # pasture AABB::contains [recalled]: reject on p < min || p > max
    return not (any(p[a] < bmin[a] for a in range(3)) or any(p[a] > bmax[a] for a in range(3)))


def lazer_section(out, last, pts_world):
    import sys
    sys.path.insert(0, os.path.dirname(HERE))
    import _lz4ref
    real = _lz4ref.load()
    if real is None:
        old = json.load(open(os.path.join(HERE, "expected.json")))
        out["lazer"] = old["lazer"]
        print("liblz4 not found: kept tiny_fmt2.lazer / lz4_frames.json")
        return
    n = len(POINTS)
    # lz4 crate EncoderBuilder defaults: 64 KiB linked blocks, content checksum, level 0
    frame = lambda b: real.compress_frame(bytes(b), 0, False, True, False, False, 0)
    nattr = 9  # format 2: 8 + colour (lazer_reader.rs:92-105)
    nb = (n + LAZER_BLOCK - 1) // LAZER_BLOCK
    head = bytearray(last[:227])
    body = bytearray(struct.pack("<Q", LAZER_BLOCK)) + bytearray(8 * nb)
    for b in range(nb):
        lo, hi = b * LAZER_BLOCK, min(n, (b + 1) * LAZER_BLOCK)
        cols = [b"".join(struct.pack("<iii", *POINTS[i][:3]) for i in range(lo, hi)),          # 0 positions
                b"".join(struct.pack("<H", 100 + i) for i in range(lo, hi)),                  # 1 intensity
                bytes([0x11] * (hi - lo)),                                                    # 2 return bits
                bytes(POINTS[i][3] for i in range(lo, hi)),                                   # 3 classification
                bytes(hi - lo), bytes(hi - lo), bytes(2 * (hi - lo)), bytes(hi - lo),         # 4..7
                b"".join(struct.pack("<HHH", *POINTS[i][4]) for i in range(lo, hi))]           # 8 colour
        blobs = [frame(c) for c in cols]
        at = 227 + len(body)
        struct.pack_into("<Q", body, 8 + 8 * b, at)
        table = bytearray(8 * nattr)
        pos = at + 8 * nattr
        for k, blob in enumerate(blobs):
            struct.pack_into("<Q", table, 8 * k, pos)
            pos += len(blob)
        body += table + b"".join(blobs)
    lazer = bytes(head) + bytes(body)
    open(os.path.join(HERE, "tiny_fmt2.lazer"), "wb").write(lazer)

    wpos = [tuple(OFFSET[a] + SCALE[a] * float(p[a]) for a in range(3)) for p in POINTS]  # lazer_reader.rs:602-609
    hmin = [min(p[a] for p in pts_world) for a in range(3)]
    hmax = [max(p[a] for p in pts_world) for a in range(3)]

    def rec(i):
        return struct.pack("<dddHHHB", *wpos[i], *POINTS[i][4], POINTS[i][3])

    queries = [((100.0, 200.0, -10.0), (110.0, 210.0, 0.0)), ((0.0, 0.0, -100.0), (1000.0, 1000.0, 100.0)),
               (wpos[3], wpos[3]), ((1e6, 1e6, 1e6), (2e6, 2e6, 2e6)),
               ((hmin[0], hmin[1], hmin[2]), (hmin[0], hmax[1], hmax[2]))]
    res = {"block_size": LAZER_BLOCK, "bounds": [], "class": []}
    for bmin, bmax in queries:
        idx = [i for i in range(n) if contains(bmin, bmax, wpos[i])] if intersects(hmin, hmax, bmin, bmax) else []
        res["bounds"].append({"min": list(bmin), "max": list(bmax), "count": len(idx), "indices": idx,
                              "points_hex": b"".join(rec(i) for i in idx).hex()})
    for cls in sorted({p[3] for p in POINTS} | {99}):
        # lazer.rs:80-116: the buffer is never cleared -> chunk k filters points [0, points_in_chunk(k)) of chunk 0
        idx = []
        for k in range(nb):
            in_chunk = min(LAZER_BLOCK, n - k * LAZER_BLOCK)
            idx += [i for i in range(in_chunk) if POINTS[i][3] == cls]
        res["class"].append({"class": cls, "count": len(idx), "indices": idx, "true_count": sum(1 for p in POINTS if p[3] == cls)})
    out["lazer"] = res

    # real-liblz4 frames as fixtures for the LZ4 readers; content = pattern * repeat
    frames = []
    patterns = [("one_byte", b"x", 1), ("hello", b"hello ", 9), ("zeros_3_blocks", b"\0", 150000), ("text_2_blocks", b"lorem ipsum dolor sit amet, ", 3000),
                ("period3", bytes([1, 2, 3]), 30000), ("bytes256", bytes(range(256)), 2)]
    for name, pat, rep in patterns:
        content = pat * rep
        for tag, kw in (("default", dict()), ("indep_bsum_size", dict(independent=True, block_checksum=True, content_size=True)),
                        ("nochecks_256k", dict(block_id=5, content_checksum=False)), ("hc", dict(level=9))):
            fr = real.compress_frame(content, **kw)
            if len(fr) > 2000:
                continue
            frames.append({"name": name + "/" + tag, "pattern": pat.hex(), "repeat": rep, "frame": fr.hex()})
    json.dump({"liblz4_version": real.version(), "frames": frames}, open(os.path.join(HERE, "lz4_frames.json"), "w"), indent=1)
    print("wrote tiny_fmt2.lazer (%d B), lz4_frames.json (%d frames)" % (len(lazer), len(frames)))


if __name__ == "__main__":
    main()
