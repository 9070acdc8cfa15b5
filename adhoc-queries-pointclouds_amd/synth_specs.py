"""Synthetic dataset recipes and the named queries of the reference's experiments.

The real navvis / doc / ca13 datasets cannot be downloaded (no network), so bench.py and the tests
use deterministic synthetic stand-ins with the same point counts and coordinate ranges
(SURVEY.md §8d).  A recipe is a list of ``SynthSpec`` (one per file); the bytes are produced either
on the host by the oracle-side generator (oracle/synth.c) or directly in HBM by
``pcq_synth_fill_dev`` — both are integer-only and bit-identical.

Named queries: query/src/bin/run_query_experiments.rs:109-144, 320-343.
Dataset sizes: query/src/bin/run_postgis_queries.rs:22-24.
"""
from __future__ import annotations

import math
from typing import List, Optional

from .binding import SynthSpec

# run_query_experiments.rs:109-144
QUERIES = {
    "navvis_S": (0, 0, 0, 2, 2, 2),
    "navvis_L": (0, 0, 0, 20, 20, 5),
    "navvis_XL": (-23.108, -21.261, -10.029, 28.588, 27.123, 5.959),
    "doc_S": (390000, 130000, 0, 390500, 140000, 200),
    "doc_L": (390000, 130000, 0, 400000, 140000, 200),
    "doc_XL": (389400, 124200, -94.88, 406200, 148200, 760.03),
    "ca13_S": (665000, 3910000, 0, 705000, 3950000, 480),
    "ca13_L": (665000, 3910000, 0, 710000, 3950000, 480),
    "ca13_XL": (643431.76, 3883547.565, -46194.145, 736910.93, 3977026.735, 47285.025),
}
CLASS_BUILDING = 6   # run_query_experiments.rs:320-331
CLASS_ABSENT = 19    # :332-343 — does not occur, 0 matches

FULL_POINTS = {"navvis": 56_200_000, "doc": 854_000_000, "ca13": 2_608_000_000}


def box(name: str):
    q = QUERIES[name]
    return list(q[:3]), list(q[3:])


def _spec(seed, n, fmt, scale, offset, lo, span, zo=None, classes=None) -> SynthSpec:
    s = SynthSpec()
    s.seed, s.n, s.format = seed, n, fmt
    for a in range(3):
        s.scale[a] = scale[a]
        s.offset[a] = offset[a]
        s.lo[a] = lo[a]
        s.span[a] = span[a]
        assert span[a] >= 1 and -(2 ** 31) <= lo[a] and lo[a] + span[a] - 1 < 2 ** 31
    if zo:
        s.zo_prob16, s.zo_lo, s.zo_span = zo
    classes = classes or [(0, 1.0)]
    assert len(classes) <= 8
    cum = 0.0
    for j, (val, p) in enumerate(classes):
        cum += p
        s.cls_val[j] = val
        s.cls_cum16[j] = min(65536, int(round(cum * 65536)))
    s.cls_cum16[len(classes) - 1] = 65536
    s.n_classes = len(classes)
    return s


def synth_navvis(points_per_file: Optional[int] = None) -> List[SynthSpec]:
    """1 file, format 2 (RGB), scale 0.001, offset 0, uniform in the navvis-XL box."""
    n = points_per_file or FULL_POINTS["navvis"]
    lo = (-23108, -21261, -10029)
    hi = (28588, 27123, 5959)
    return [_spec(0x4E415601, n, 2, (0.001,) * 3, (0.0,) * 3, lo, tuple(h - l + 1 for l, h in zip(lo, hi)),
                  classes=[(1, 0.6), (2, 0.3), (6, 0.1)])]


def synth_doc(points_per_file: Optional[int] = None, files: int = 8) -> List[SynthSpec]:
    """8 files (x strips of the doc-XL box), format 1, scale 0.01, offset = doc-XL min corner."""
    n = points_per_file or FULL_POINTS["doc"] // 8
    off = (389400.0, 124200.0, -94.88)
    strip = 1_680_000 // 8
    pmf = [(1, 0.30), (2, 0.45), (5, 0.15), (6, 0.08), (7, 0.01), (9, 0.01)]
    out = []
    for f in range(files):
        out.append(_spec(0x444F4301 + f, n, 1, (0.01,) * 3, off, (strip * (f % 8), 0, 0), (strip, 2_400_001, 85_492),
                         classes=pmf))
    return out


def synth_ca13(points_per_file: Optional[int] = None, files: int = 16) -> List[SynthSpec]:
    """16 files = 4x4 x/y tiles of the ca13-XL box, format 1, scale 0.01, offset 0; z uniform in
    [0, 480] m for 99 % of the points and in the whole XL z range for 1 %."""
    n = points_per_file or FULL_POINTS["ca13"] // 16
    x_lo, x_hi = 64343176, 73691093
    y_lo, y_hi = 388354757, 397702673
    z_all_lo, z_all_hi = -4619414, 4728502
    wx = math.ceil((x_hi - x_lo + 1) / 4)
    wy = math.ceil((y_hi - y_lo + 1) / 4)
    out = []
    for f in range(files):
        tx, ty = (f % 16) % 4, (f % 16) // 4
        lx, ly = x_lo + tx * wx, y_lo + ty * wy
        sx, sy = min(wx, x_hi + 1 - lx), min(wy, y_hi + 1 - ly)
        out.append(_spec(0xCA130001 + f, n, 1, (0.01,) * 3, (0.0,) * 3, (lx, ly, 0), (sx, sy, 48001),
                         zo=(655, z_all_lo, z_all_hi - z_all_lo + 1),
                         classes=[(1, 0.25), (2, 0.55), (5, 0.10), (6, 0.05), (9, 0.05)]))
    return out


def dataset(name: str, points_per_file: Optional[int] = None, files: Optional[int] = None) -> List[SynthSpec]:
    if name == "navvis":
        return synth_navvis(points_per_file)
    if name == "doc":
        return synth_doc(points_per_file, files or 8)
    if name == "ca13":
        return synth_ca13(points_per_file, files or 16)
    raise KeyError(name)


def header_fields(spec: SynthSpec) -> dict:
    """The LAS header fields the query path reads (number of points, scale, offset, min/max bounds)
    for a synthetic file, computed exactly as the generator writes them: extreme integer
    coordinates mapped with `(i as f64 * scale) + offset` (last.rs:156-160).  Python floats are IEEE
    doubles, so these equal the bytes in the generated header (tests/test_synth_and_sharding.py::test_header_fields_equal_the_generated_header)."""
    mn, mx = [], []
    for a in range(3):
        lo = spec.lo[a]
        hi = spec.lo[a] + spec.span[a] - 1
        if a == 2 and spec.zo_prob16 > 0:
            lo = min(lo, spec.zo_lo)
            hi = max(hi, spec.zo_lo + spec.zo_span - 1)
        a0 = (float(lo) * spec.scale[a]) + spec.offset[a]
        a1 = (float(hi) * spec.scale[a]) + spec.offset[a]
        mn.append(min(a0, a1))
        mx.append(max(a0, a1))
    return {"n": int(spec.n), "format": int(spec.format), "scale": list(spec.scale), "offset": list(spec.offset),
            "min": mn, "max": mx}


def aabb_intersects(amin, amax, bmin, bmax) -> bool:
    """pasture AABB::intersects (inclusive) — the file-level early-out of last.rs:92-94."""
    return all(amin[a] <= bmax[a] and amax[a] >= bmin[a] for a in range(3))
