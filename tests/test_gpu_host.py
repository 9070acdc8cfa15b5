"""GPU parity at the reference's operator level: Searcher::search_file with each collector, on LAST and
LAS files, through libpcq_query.so (the C view of the C++ host layer) — and the `query` CLI against the
oracle CLI.  The scans run in libpcq.so (HIP); the oracle is only the checker.
"""
import ctypes as C
import importlib
import json
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "adhoc-queries-pointclouds_amd")
G = json.load(open(os.path.join(HERE, "golden", "expected.json")))

pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
POINT_DTYPE = pkg.POINT_DTYPE


class Q:
    """ctypes view of include/pcq_query.h"""

    def __init__(self):
        lib = self.lib = C.CDLL(os.path.join(PKG, "libpcq_query.so"))
        vp, P, u64 = C.c_void_p, C.POINTER, C.c_uint64
        dd = P(C.c_double)
        lib.pcq_query_last_error.restype = C.c_char_p
        lib.pcq_query_collector_new_count.argtypes = [C.c_int, P(vp)]
        lib.pcq_query_collector_new_buffer.argtypes = [C.c_int, P(vp)]
        lib.pcq_query_collector_new_grid.argtypes = [C.c_int, dd, dd, C.c_double, P(vp)]
        lib.pcq_query_collector_free.argtypes = [vp]
        lib.pcq_query_collector_point_count.argtypes = [vp, P(u64)]
        lib.pcq_query_collector_has_points.argtypes = [vp]
        lib.pcq_query_collector_points.argtypes = [vp, vp, u64, P(u64)]
        lib.pcq_query_collector_grid_cells.argtypes = [vp, vp, u64, P(u64)]
        lib.pcq_query_search_file_bounds.argtypes = [C.c_char_p, dd, dd, C.c_int, vp, P(C.c_int)]
        lib.pcq_query_search_file_class.argtypes = [C.c_char_p, C.c_uint8, C.c_int, vp]
        lib.pcq_query_test_plan_replace_execute.argtypes = [C.c_char_p, C.c_char_p, dd, dd, vp]

    @staticmethod
    def d3(v):
        return (C.c_double * 3)(*[float(x) for x in v])

    def collector(self, kind, bmin=None, bmax=None, cell=None):
        h = C.c_void_p()
        if kind == "count":
            rc = self.lib.pcq_query_collector_new_count(0, C.byref(h))
        elif kind == "buffer":
            rc = self.lib.pcq_query_collector_new_buffer(0, C.byref(h))
        else:
            rc = self.lib.pcq_query_collector_new_grid(0, self.d3(bmin), self.d3(bmax), cell, C.byref(h))
        assert rc == 0, self.lib.pcq_query_last_error()
        return h

    def count(self, h):
        n = C.c_uint64()
        assert self.lib.pcq_query_collector_point_count(h, C.byref(n)) == 0, self.lib.pcq_query_last_error()
        return n.value

    def points(self, h):
        n = C.c_uint64()
        assert self.lib.pcq_query_collector_points(h, None, 0, C.byref(n)) == 0
        out = np.zeros(n.value, dtype=POINT_DTYPE)
        if n.value:
            assert self.lib.pcq_query_collector_points(h, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)) == 0
        return out

    def cells(self, h):
        n = C.c_uint64()
        assert self.lib.pcq_query_collector_grid_cells(h, None, 0, C.byref(n)) == 0
        out = np.zeros(n.value, dtype=np.uint64)
        if n.value:
            assert self.lib.pcq_query_collector_grid_cells(h, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)) == 0
        return out

    def search_bounds(self, path, bmin, bmax, h, optimized=1):
        rec = C.c_int(-1)
        rc = self.lib.pcq_query_search_file_bounds(path.encode(), self.d3(bmin), self.d3(bmax), optimized, h, C.byref(rec))
        return rc, rec.value

    def search_class(self, path, cls, h, optimized=1):
        return self.lib.pcq_query_search_file_class(path.encode(), cls, optimized, h)

    def free(self, h):
        self.lib.pcq_query_collector_free(h)


@pytest.fixture(scope="module")
def q():
    return Q()


@pytest.fixture(scope="module")
def files(oracle, tmp_path_factory):
    """A small multi-format dataset on disk: LAST and LAS of formats 0-3, anisotropic scales."""
    d = tmp_path_factory.mktemp("data")
    out = []
    for fmt in (0, 1, 2, 3):
        spec = specs._spec(9000 + fmt, 70_001 + 13 * fmt, fmt, (0.01, 0.02, 0.05), (100.0, -200.0, 7.5), (-5000, -5000, -1000),
                           (10001, 10001, 2001), classes=[(1, 0.4), (2, 0.3), (6, 0.2), (134, 0.1)])
        for ext in ("last", "las"):
            p = str(d / f"f{fmt}.{ext}")
            oracle.synth_write(spec, p)
            out.append(p)
    return out


BOXES = [((90.0, -250.0, 0.0), (120.0, -150.0, 20.0)), ((0.0, -400.0, -100.0), (200.0, 0.0, 100.0)),
         ((149.99, -400.0, -100.0), (150.0, 0.0, 100.0)), ((500.0, 500.0, 500.0), (600.0, 600.0, 600.0))]


def test_search_file_bounds_count_buffer(oracle, q, files):
    for path in files:
        for bmin, bmax in BOXES:
            oc, ob = oracle.count_collector(), oracle.buffer_collector()
            rc_o, rec_o = oracle.search_file(path, 0, bmin, bmax, 0, oc)
            oracle.search_file(path, 0, bmin, bmax, 0, ob)
            hc, hb = q.collector("count"), q.collector("buffer")
            rc_c, rec_c = q.search_bounds(path, bmin, bmax, hc)
            rc_b, _ = q.search_bounds(path, bmin, bmax, hb)
            assert rc_c == rc_o == rc_b, (path, q.lib.pcq_query_last_error())
            assert rec_c == rec_o  # "Point record size" side output (las.rs:73)
            assert q.count(hc) == oc.point_count(), (path, bmin)
            assert q.points(hb).tobytes() == ob.points().tobytes(), (path, bmin)
            q.free(hc), q.free(hb), oc.free(), ob.free()


def test_search_file_class_count_buffer(oracle, q, files):
    for path in files:
        for cls in (6, 134, 19):
            oc, ob = oracle.count_collector(), oracle.buffer_collector()
            assert oracle.search_file(path, 1, None, None, cls, oc)[0] == 0
            oracle.search_file(path, 1, None, None, cls, ob)
            hc, hb = q.collector("count"), q.collector("buffer")
            assert q.search_class(path, cls, hc) == 0 and q.search_class(path, cls, hb) == 0
            assert q.count(hc) == oc.point_count()
            assert q.points(hb).tobytes() == ob.points().tobytes()
            q.free(hc), q.free(hb), oc.free(), ob.free()


@pytest.mark.parametrize("cell", [0.5, 3.0, 12.5])
def test_search_file_grid_per_file_and_sequential(oracle, q, files, cell):
    bmin, bmax = BOXES[1]
    # --parallel: one grid per file (main.rs:156)
    for path in files[:4]:
        og = oracle.grid_collector(bmin, bmax, cell)
        assert oracle.search_file(path, 0, bmin, bmax, 0, og)[0] == 0
        hg = q.collector("grid", bmin, bmax, cell)
        assert q.search_bounds(path, bmin, bmax, hg)[0] == 0
        keys, pts = q.cells(hg), q.points(hg)
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(keys[order], og.grid_cells())
        assert pts[order].tobytes() == og.points().tobytes()
        q.free(hg), og.free()
    # sequential: ONE grid across all files, in file order (main.rs:129-133) — first seen wins across files
    og = oracle.grid_collector(bmin, bmax, cell)
    hg = q.collector("grid", bmin, bmax, cell)
    for path in files:
        assert oracle.search_file(path, 0, bmin, bmax, 0, og)[0] == 0
        assert q.search_bounds(path, bmin, bmax, hg)[0] == 0
    keys, pts = q.cells(hg), q.points(hg)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(keys[order], og.grid_cells())
    assert pts[order].tobytes() == og.points().tobytes()
    q.free(hg), og.free()


def test_class_query_with_grid_collector(oracle, q, files):
    path = files[4]  # f2.last
    bmin, bmax = (50.0, -300.0, -50.0), (150.0, -100.0, 60.0)  # any grid box (main.rs:255-259 uses the header union)
    og = oracle.grid_collector(bmin, bmax, 4.0)
    assert oracle.search_file(path, 1, None, None, 6, og)[0] == 0
    hg = q.collector("grid", bmin, bmax, 4.0)
    assert q.search_class(path, 6, hg) == 0
    keys, pts = q.cells(hg), q.points(hg)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(keys[order], og.grid_cells())
    assert pts[order].tobytes() == og.points().tobytes()
    q.free(hg), og.free()


def _records(pts):
    return [[float(p["x"]).hex(), float(p["y"]).hex(), float(p["z"]).hex(), int(p["r"]), int(p["g"]), int(p["b"]),
             int(p["classification"])] for p in pts]


def test_golden_tiny_files_through_the_product(q):
    """The committed known-answer vectors, directly against the HIP path (no oracle involved)."""
    for fname in ("tiny_fmt2.last", "tiny_fmt3.las"):
        path = os.path.join(HERE, "golden", fname)
        for name, exp in G["tiny"]["bounds"].items():
            hc, hb = q.collector("count"), q.collector("buffer")
            rc, _ = q.search_bounds(path, exp["bmin"], exp["bmax"], hc)
            rc2, _ = q.search_bounds(path, exp["bmin"], exp["bmax"], hb)
            if exp["panic"]:
                assert rc == rc2 == -7 and q.lib.pcq_query_last_was_panic() == 1
            else:
                assert rc == rc2 == 0
                assert q.count(hc) == len(exp["indices"]), (fname, name)
                assert _records(q.points(hb)) == [list(r) for r in exp["records"]], (fname, name)
            q.free(hc), q.free(hb)
        for cls, exp in G["tiny"]["class"].items():
            hc, hb = q.collector("count"), q.collector("buffer")
            assert q.search_class(path, int(cls), hc) == 0 and q.search_class(path, int(cls), hb) == 0
            assert q.count(hc) == len(exp["indices"])
            assert _records(q.points(hb)) == [list(r) for r in exp["records"]]
            q.free(hc), q.free(hb)
    path = os.path.join(HERE, "golden", "tiny_fmt2.last")
    all_recs = G["tiny"]["bounds"]["box_everything"]["records"]
    for name, exp in G["tiny"]["grid"].items():
        qq = G["tiny"]["bounds"][exp["query"]]
        hg = q.collector("grid", qq["bmin"], qq["bmax"], exp["cell"])
        assert q.search_bounds(path, qq["bmin"], qq["bmax"], hg)[0] == 0
        keys, pts = q.cells(hg), q.points(hg)
        order = np.argsort(keys, kind="stable")
        assert [int(k) for k in keys[order]] == exp["keys"], name  # includes the mask-aliasing case (cell 4 -> key 0)
        assert _records(pts[order]) == [list(all_recs[i]) for i in exp["winners"]], name
        q.free(hg)


def test_golden_grid_traps_through_the_product(q, tmp_path):
    """grid_sampling.rs tests + alias / tie traps: points are written as a LAST file (scale 1e-3) and
    run through a class query so that every point reaches the collector in file order."""
    for name, exp in G["grid"].items():
        if not name.startswith("lat_"):
            continue  # off-lattice vectors are pinned at the oracle level (test_oracle_golden.py)
        pts = exp["points"]
        n = len(pts)
        img = bytearray(227 + 20 * n)
        img[0:4] = b"LASF"
        img[24], img[25] = 1, 2
        struct.pack_into("<H", img, 94, 227)
        struct.pack_into("<I", img, 96, 227)
        img[104] = 0
        struct.pack_into("<H", img, 105, 20)
        struct.pack_into("<I", img, 107, n)
        struct.pack_into("<ddd", img, 131, 1 / 64, 1 / 64, 1 / 64)  # exactly representable: positions rebuild exactly
        struct.pack_into("<ddd", img, 155, 0.0, 0.0, 0.0)
        for a in range(3):
            struct.pack_into("<dd", img, 179 + 16 * a, 100.0, -100.0)
        for i, p in enumerate(pts):
            ints = [v * 64 for v in p]
            assert all(v == round(v) for v in ints)
            struct.pack_into("<iii", img, 227 + 12 * i, *[int(v) for v in ints])
            img[227 + 15 * n + i] = 7
        path = str(tmp_path / f"{name}.last")
        open(path, "wb").write(bytes(img))
        hg = q.collector("grid", exp["bmin"], exp["bmax"], exp["cell"])
        assert q.search_class(path, 7, hg) == 0
        keys, gp = q.cells(hg), q.points(hg)
        order = np.argsort(keys, kind="stable")
        assert [int(k) for k in keys[order]] == exp["keys"], name
        got = [(float(p["x"]), float(p["y"]), float(p["z"])) for p in gp[order]]
        assert got == [tuple(pts[i]) for i in exp["winners"]], name
        q.free(hg)


def test_errors_match_the_reference_behaviour(oracle, q, files, tmp_path):
    good = files[0]
    data = open(good, "rb").read()
    # truncated positions block -> UnexpectedEof
    p = str(tmp_path / "trunc.last")
    open(p, "wb").write(data[:227 + 1000])
    h = q.collector("count")
    assert q.search_bounds(p, *BOXES[1], h)[0] == -5
    assert oracle.search_file(p, 0, *BOXES[1], 0, oracle.count_collector())[0] == -5
    # bad signature / short header
    p2 = str(tmp_path / "sig.last")
    open(p2, "wb").write(b"XXXX" + data[4:])
    assert q.search_bounds(p2, *BOXES[1], h)[0] == -2
    p3 = str(tmp_path / "short.las")
    open(p3, "wb").write(data[:50])
    assert q.search_class(p3, 6, h) == -2
    # extensions (searcher.rs:84-88)
    p4 = str(tmp_path / "x.xyz")
    open(p4, "wb").write(data)
    assert q.search_bounds(p4, *BOXES[1], h)[0] == -4
    assert q.search_class(str(tmp_path / "noext"), 6, h) == -4
    # missing file
    assert q.search_bounds(str(tmp_path / "missing.last"), *BOXES[1], h)[0] == -1
    # the non-optimized implementation and compressed formats are outside the hot path: loud error, no fallback
    assert q.search_bounds(good, *BOXES[1], h, optimized=0)[0] == -11
    p5 = str(tmp_path / "c.laz")
    open(p5, "wb").write(data)
    assert q.search_bounds(p5, *BOXES[1], h)[0] == -11
    # format byte with the compressed bit: bounds path fails in the header, class path masks it (last.rs:222)
    flagged = bytearray(data)
    flagged[104] |= 0x80
    p6 = str(tmp_path / "flag.last")
    open(p6, "wb").write(bytes(flagged))
    assert q.search_bounds(p6, *BOXES[1], h)[0] == -2
    hc = q.collector("count")
    oc = oracle.count_collector()
    assert q.search_class(p6, 6, hc) == 0 and oracle.search_file(p6, 1, None, None, 6, oc)[0] == 0
    assert q.count(hc) == oc.point_count()
    q.free(h), q.free(hc)


def _cli(exe, args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([exe] + args, capture_output=True, text=True, env=e)
    lines = r.stdout.splitlines()
    body = [l for l in lines if not l.startswith("Searched ")]
    timing = [l for l in lines if l.startswith("Searched ")]
    return r.returncode, body, timing, r.stderr


QUERY = os.path.join(PKG, "host", "query")
ORACLE_CLI = os.path.join(ROOT, "oracle", "query_oracle")


def _cli_hooks(args, device_slots=None, allreduce_fail=0, env=None):
    """The CLI through the test entry of the C view (pcq_query_main_with_hooks): a process of its own, like the binary, with
    the parallel driver's test hooks — which the `query` binary cannot reach (no environment variable sets them)."""
    code = ("import ctypes as C, sys\n"
            "lib = C.CDLL(%r)\n"
            "argv = ['query'] + sys.argv[3:]\n"
            "arr = (C.c_char_p * len(argv))(*[a.encode() for a in argv])\n"
            "slots = sys.argv[1].encode() if sys.argv[1] != '-' else None\n"
            "rc = lib.pcq_query_main_with_hooks(len(argv), arr, slots, int(sys.argv[2]))\n"
            "sys.stdout.flush()\n"
            "sys.exit(rc)\n") % os.path.join(PKG, "libpcq_query.so")
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, "-c", code, device_slots or "-", str(allreduce_fail)] + args, capture_output=True, text=True, env=e)
    lines = r.stdout.splitlines()
    return r.returncode, [l for l in lines if not l.startswith("Searched ")], [l for l in lines if l.startswith("Searched ")], r.stderr


@pytest.mark.parametrize("mode", [["--parallel"], []])
@pytest.mark.parametrize("query_args", [["--bounds", "90;-250;0;120;-150;20"], ["--bounds", "0;-400;-100;200;0;100"],
                                        ["--class", "6"], ["--class", "19"], ["--bounds", "500;500;500;600;600;600"],
                                        ["--bounds", "0;-400;-100;200;0;100", "--density", "5"], ["--class", "6", "--density", "2.5"]])
def test_cli_stdout_matches_oracle_cli(files, mode, query_args):
    d = os.path.dirname(files[0])
    args = ["-i", d, "--optimized"] + mode + query_args
    rc_p, body_p, timing_p, err_p = _cli(QUERY, args)
    rc_o, body_o, timing_o, err_o = _cli(ORACLE_CLI, args)
    assert rc_p == rc_o == 0, (err_p, err_o)
    # "Point record size" lines are printed from worker threads in the reference: compare as a multiset
    assert sorted(body_p) == sorted(body_o)
    assert body_p[0] == body_o[0] == "Searching 8 files..."
    assert len(timing_p) == 1 and timing_p[0].split(" MiB in ")[0] == timing_o[0].split(" MiB in ")[0]
    # small chunks: the host streaming pipeline must not change any result
    rc_c, body_c, _, _ = _cli(QUERY, args, env={"PCQ_CHUNK_POINTS": "4096"})
    assert rc_c == 0 and sorted(body_c) == sorted(body_o)


@pytest.mark.parametrize("box", ["nan;0;0;1;1;1", "0;-400;-100;inf;inf;inf", "-inf;-inf;-inf;inf;inf;inf", "0;0;0;nan;nan;nan",
                                 "-1e308;-1e308;-1e308;1e308;1e308;1e308", "1e-320;-400;-100;200;0;100", "90;-250;0;90;-250;0"])
def test_cli_non_finite_and_extreme_boxes(files, box):
    """`str::parse::<f64>` accepts nan / inf; NaN compares false everywhere (no panic in from_min_max, no
    intersection), infinities saturate in `as i64` (last.rs:98-109)."""
    d = os.path.dirname(files[0])
    args = ["-i", d, "--optimized", "--parallel", "--bounds", box]
    rc_p, body_p, _, err_p = _cli(QUERY, args)
    rc_o, body_o, _, err_o = _cli(ORACLE_CLI, args)
    assert rc_p == rc_o, (box, err_p, err_o)
    assert sorted(body_p) == sorted(body_o), box


def test_cli_density_with_output_dir_writes_one_file_per_grid(files, tmp_path):
    d = os.path.dirname(files[0])
    for mode in (["--parallel"], []):
        out = tmp_path / ("o" + str(len(mode)))
        out.mkdir()
        args = ["-i", d, "--optimized", "--bounds", "0;-400;-100;200;0;100", "--density", "7.5", "-o", str(out)] + mode
        rc_p, body_p, _, err_p = _cli(QUERY, args)
        rc_o, body_o, _, _ = _cli(ORACLE_CLI, args)
        assert rc_p == rc_o == 0, err_p
        assert sorted(body_p) == sorted(body_o)  # "Writing N points" per grid (per file in --parallel, one otherwise)
        written = sorted(os.listdir(out))
        assert written == [f"matching_points_{i}.las" for i in range(len(written))]
        assert len(written) == len([l for l in body_p if l.startswith("Writing ")])


def test_cli_extra_flags_do_not_change_stdout(files, tmp_path):
    d = os.path.dirname(files[0])
    base = ["-i", d, "--optimized", "--parallel", "--bounds", "0;-400;-100;200;0;100"]
    rc0, body0, _, _ = _cli(QUERY, base)
    sj = tmp_path / "stats.json"
    rc1, body1, _, err = _cli(QUERY, base + ["--gpus", "1", "--threads-per-gpu", "3", "--stats-json", str(sj)])
    assert rc0 == rc1 == 0, err
    assert sorted(body0) == sorted(body1)
    st = json.load(open(sj))
    assert st["files"] == 8 and len(st["per_file"]) == 8 and st["gpus"] == 1 and st["parallel"] is True
    assert sorted(e["path"] for e in st["per_file"]) == sorted(files)


def test_cli_single_file_and_output_dir(oracle, files, tmp_path):
    path = [f for f in files if f.endswith("f2.last")][0]
    out = tmp_path / "out"
    out.mkdir()
    rc, body, _, err = _cli(QUERY, ["-i", path, "--bounds", "90;-250;0;120;-150;20", "--optimized", "--parallel", "-o", str(out)])
    assert rc == 0, err
    ob = oracle.buffer_collector()
    oracle.search_file(path, 0, (90.0, -250.0, 0.0), (120.0, -150.0, 20.0), 0, ob)
    want = ob.points()
    assert body == ["Searching 1 files...", f"Writing {len(want)} points"]
    # decoded-record parity of the written LAS (dump_points.rs:63-116): version 1.2, format 2,
    # offset = min position, scale rule, same class / RGB, positions within half a scale unit
    data = open(out / "matching_points_0.las", "rb").read()
    assert data[:4] == b"LASF" and data[24] == 1 and data[25] == 2 and data[104] == 2
    n = struct.unpack_from("<I", data, 107)[0]
    assert n == len(want) and struct.unpack_from("<H", data, 105)[0] == 26
    scale = struct.unpack_from("<ddd", data, 131)
    offset = struct.unpack_from("<ddd", data, 155)
    assert offset == (want["x"].min(), want["y"].min(), want["z"].min())
    ext = max(want["x"].max() - want["x"].min(), want["y"].max() - want["y"].min(), want["z"].max() - want["z"].min())
    assert scale[0] == scale[1] == scale[2] == max(0.001, 10.0 ** np.ceil(np.log10(ext / 2147483647.0)))
    rec = np.frombuffer(data[227:227 + 26 * n], dtype=np.dtype([("xyz", "<i4", 3), ("i", "<u2"), ("bits", "u1"), ("cls", "u1"),
                                                                 ("rest", "u1", 4), ("rgb", "<u2", 3)]))
    assert np.array_equal(rec["cls"], want["classification"])
    assert np.array_equal(rec["rgb"], np.stack([want["r"], want["g"], want["b"]], axis=1))
    for a, k in enumerate("xyz"):
        assert np.all(np.abs(rec["xyz"][:, a] * scale[a] + offset[a] - want[k]) <= 0.5 * scale[a] + 1e-9)


def test_cli_query_resolved_on_the_host_never_wakes_the_gpu(files):
    """Host first (run_search.cpp): the header prologue of every file runs before any GPU context exists.  A box that
    misses every file's header AABB (last.rs:92-94) is answered without HIP start-up — no context is created — while a
    box that hits at least one file creates exactly one (one GPU, one host thread per GPU)."""
    d = os.path.dirname(files[0])
    miss = ["-i", d, "--optimized", "--parallel", "--bounds", "500;500;500;600;600;600"]
    rc_p, body_p, _, err_p = _cli(QUERY, miss, env={"PCQ_TIMING": "1"})
    rc_o, body_o, _, _ = _cli(ORACLE_CLI, miss)
    assert rc_p == rc_o == 0 and sorted(body_p) == sorted(body_o)
    assert "Found 0 matching points" in body_p
    assert "0 of 8 files need the GPU" in err_p and "context on device" not in err_p
    hit = ["-i", d, "--optimized", "--parallel", "--bounds", "90;-250;0;120;-150;20"]
    rc_p, body_p, _, err_p = _cli(QUERY, hit, env={"PCQ_TIMING": "1"})
    rc_o, body_o, _, _ = _cli(ORACLE_CLI, hit)
    assert rc_p == rc_o == 0 and sorted(body_p) == sorted(body_o)
    assert err_p.count("context on device") == 1
    # the same with an output directory and with a density: collectors that yield points
    for extra in (["--density", "5"],):
        rc_p, body_p, _, err_p = _cli(QUERY, miss + extra, env={"PCQ_TIMING": "1"})
        rc_o, body_o, _, _ = _cli(ORACLE_CLI, miss + extra)
        assert rc_p == rc_o == 0 and sorted(body_p) == sorted(body_o) and "context on device" not in err_p


@pytest.mark.parametrize("when", ["early", "late", "group"])
def test_cli_count_merge_survives_a_failing_allreduce(files, when):
    """main.rs:164-180 on several GPUs is one RCCL all-reduce of the per-GPU counters.  When the collective fails — before
    it touched anything, or after the reduction had run on every rank (a late stream error) — the CLI sums the per-GPU
    counts on the host: the all-reduce is out of place, so those counts are never a partially reduced value, and the
    printed count is the oracle's either way (with the warning on stderr).  The failure is injected into the real RCCL
    call path (one-rank communicator; several GPUs are not available to the tests)."""
    d = os.path.dirname(files[0])
    for query_args in (["--bounds", "0;-400;-100;200;0;100"], ["--class", "6"]):
        args = ["-i", d, "--optimized", "--parallel"] + query_args
        rc_p, body_p, _, err_p = _cli_hooks(args, allreduce_fail={"early": 1, "late": 2, "group": 3}[when])
        rc_o, body_o, _, _ = _cli(ORACLE_CLI, args)
        assert rc_p == rc_o == 0, err_p
        assert sorted(body_p) == sorted(body_o)
        assert "all-reduce of the per-GPU counts failed" in err_p and "injected failure" in err_p and "summing on the host" in err_p
        assert {"early": "before the reduction", "late": "after the reduction", "group": "inside the group"}[when] in err_p


@pytest.mark.parametrize("merge", ["auto", "host", "rccl"])
def test_cli_two_device_slots_merge_like_two_gpus(files, merge, tmp_path):
    """The N > 1 paths of run_search_parallel on the one GPU the tests have: device slots "0,0" (the test entry pcq_query_main_with_hooks) give two device slots
    (own workers, own contexts, own two-word counter block each).  Count queries: the short query sums the two counter blocks on
    the host (auto, host); PCQ_MERGE=rccl asks for the all-reduce, RCCL refuses a communicator over a repeated device — a real
    failure of the real library, not an injected one — and the host sum answers with the warning.  Queries whose collectors
    yield points merge per file and need no collective.  stdout is the oracle's in every case."""
    d = os.path.dirname(files[0])
    env = {"PCQ_TIMING": "1"}
    if merge != "auto":
        env["PCQ_MERGE"] = merge
    for query_args in (["--bounds", "0;-400;-100;200;0;100"], ["--class", "6"], ["--bounds", "0;-400;-100;200;0;100", "--density", "5"]):
        args = ["-i", d, "--optimized", "--parallel"] + query_args
        rc_p, body_p, _, err_p = _cli_hooks(args + ["--threads-per-gpu", "2"], device_slots="0,0", env=env)
        rc_o, body_o, _, _ = _cli(ORACLE_CLI, args)
        assert rc_p == rc_o == 0, err_p
        assert sorted(body_p) == sorted(body_o)
        assert err_p.count("context on device 0 ready") >= 2  # both slots worked
        if "--density" in query_args:
            assert "count merge" not in err_p
        elif merge == "rccl":
            assert "count merge: RCCL all-reduce" in err_p and "all-reduce of the per-GPU counts failed" in err_p and "summing on the host" in err_p
        else:
            assert "count merge: host sum of the per-GPU counts" in err_p and "warning" not in err_p
    if merge == "auto":  # -o: the files written from two slots are the files written from one, byte for byte
        outs = []
        for slots in ("0,0", "0"):
            out = tmp_path / ("o" + str(len(slots)))
            out.mkdir()
            args = ["-i", d, "--optimized", "--parallel", "--bounds", "0;-400;-100;200;0;100", "-o", str(out), "--threads-per-gpu", "2"]
            rc, body, _, err = _cli_hooks(args, device_slots=slots)
            assert rc == 0, err
            outs.append((sorted(body), {f: open(out / f, "rb").read() for f in sorted(os.listdir(out))}))
        assert outs[0] == outs[1] and len(outs[0][1]) > 0


def test_cli_fast_exit_changes_nothing_observable(files, tmp_path):
    """The `query` binary leaves through _exit once its answer is flushed (PCQ_EXIT=fast; the default unless a profiler or a
    sanitizer hooks the end of the process) instead of releasing its contexts and running the HIP runtime's teardown: same exit
    code, same stdout, same stderr, same files as the normal return path (PCQ_EXIT=full) — also for a query that fails."""
    d = os.path.dirname(files[0])
    seen = {}
    for mode in ("fast", "full"):
        out = tmp_path / mode
        out.mkdir()
        runs = []
        for args in (["-i", d, "--optimized", "--parallel", "--bounds", "0;-400;-100;200;0;100"],
                     ["-i", d, "--optimized", "--bounds", "0;-400;-100;200;0;100", "--density", "5"],
                     ["-i", d, "--optimized", "--parallel", "--class", "6", "-o", str(out)],
                     ["-i", d, "--optimized", "--parallel", "--bounds", "not;a;box"],
                     ["-i", str(tmp_path / "missing"), "--optimized", "--parallel", "--class", "6"]):
            rc, body, _, err = _cli(QUERY, args, env={"PCQ_EXIT": mode})
            runs.append((rc, sorted(body), err.replace(str(out), "OUT")))
        seen[mode] = (runs, {f: open(out / f, "rb").read() for f in sorted(os.listdir(out))})
    assert seen["fast"] == seen["full"]
    assert seen["fast"][0][0][0] == 0 and seen["fast"][0][3][0] != 0 and len(seen["fast"][1]) > 0


def test_cli_density_over_many_small_files_takes_two_threads_per_gpu(oracle, tmp_path):
    """A --density query folds every file's grid when the file is done (a synchronisation); over 32 or more small files per GPU
    the driver feeds the GPU from two host threads unless --threads-per-gpu says otherwise (profiles/r03_density_threads.log).
    Same stdout as the oracle CLI either way; a count query over the same files keeps its single thread."""
    d = tmp_path / "small"
    d.mkdir()
    for k in range(40):
        spec = specs._spec(81000 + k, 3000, 2, (0.01, 0.01, 0.01), (0.0, 0.0, 0.0), (-5000, -5000, -1000), (10001, 10001, 2001), zo=None,
                           classes=[(1, 0.5), (6, 0.5)])
        oracle.synth_image(spec, transposed=True).tofile(str(d / f"f{k:02d}.last"))
    base = ["-i", str(d), "--optimized", "--parallel", "--bounds", "-20;-20;-5;20;20;5"]
    want = {}
    for name, extra in (("density", ["--density", "2.5"]), ("count", [])):
        rc_o, body_o, _, _ = _cli(ORACLE_CLI, base + extra)
        assert rc_o == 0
        want[name] = sorted(body_o)
    for name, extra, flags, contexts in (("density", ["--density", "2.5"], [], 2), ("density", ["--density", "2.5"], ["--threads-per-gpu", "1"], 1),
                                         ("count", [], [], 1)):
        rc, body, _, err = _cli(QUERY, base + extra + flags, env={"PCQ_TIMING": "1"})
        assert rc == 0, err
        assert sorted(body) == want[name]
        assert err.count("context on device 0 ready") == contexts, (name, flags, err)


def test_a_plan_is_not_executed_on_another_file_under_the_same_name(oracle, q, tmp_path):
    """run_search_parallel plans every file (header, offsets, box) before the first worker has a context, and the scan opens the
    path again: a file replaced in between — another inode, or rewritten in place — must not be scanned with the first one's
    offsets, scale and point count.  The test entry makes the plan, renames another file over the path, executes the plan."""
    import shutil
    spec_a = specs._spec(7001, 50_000, 1, (0.01, 0.01, 0.01), (0.0, 0.0, 0.0), (-5000, -5000, -1000), (10001, 10001, 2001))
    spec_b = specs._spec(7002, 50_000, 1, (0.02, 0.02, 0.02), (5.0, 5.0, 5.0), (-5000, -5000, -1000), (10001, 10001, 2001))  # same size, other header
    a, b = str(tmp_path / "a.last"), str(tmp_path / "b.last")
    oracle.synth_write(spec_a, a)
    oracle.synth_write(spec_b, b)
    bmin, bmax = (-20.0, -20.0, -5.0), (20.0, 20.0, 5.0)
    oc = oracle.count_collector()
    assert oracle.search_last_bounds(np.fromfile(a, dtype=np.uint8), bmin, bmax, oc) == 0
    want = oc.point_count()
    oc.free()
    assert want > 0
    h = q.collector("count")
    try:
        # nothing in between: the plan's own file
        assert q.lib.pcq_query_test_plan_replace_execute(a.encode(), None, q.d3(bmin), q.d3(bmax), h) == 0, q.lib.pcq_query_last_error()
        assert q.count(h) == want
        # another file renamed over the path (another inode)
        shutil.copy(b, str(tmp_path / "b_copy.last"))
        rc = q.lib.pcq_query_test_plan_replace_execute(a.encode(), str(tmp_path / "b_copy.last").encode(), q.d3(bmin), q.d3(bmax), h)
        assert rc != 0 and b"changed while the query was running" in q.lib.pcq_query_last_error()
        assert q.count(h) == want  # (nothing was added)
    finally:
        q.free(h)


def test_cli_more_files_than_descriptors(oracle, tmp_path):
    """run_search_parallel plans every file before the first worker starts; a plan keeps the header's values, not the open
    file (the reference opens inside the rayon task: at most one file per thread is open at a time).  1100 small LAST files
    under a descriptor limit of 256: same stdout as the oracle CLI."""
    import resource
    n_files, n = 1100, 41
    d = tmp_path / "many"
    d.mkdir()
    for k in range(n_files):
        spec = specs._spec(70000 + k, n, 1, (0.01, 0.01, 0.01), (0.0, 0.0, 0.0), (-5000, -5000, -1000), (10001, 10001, 2001), zo=None,
                           classes=[(1, 0.5), (6, 0.5)])
        oracle.synth_image(spec, transposed=True).tofile(str(d / f"f{k:04d}.last"))
    args = ["-i", str(d), "--optimized", "--parallel", "--bounds", "-20;-20;-5;20;20;5"]

    def limited():
        resource.setrlimit(resource.RLIMIT_NOFILE, (256, 256))

    r = subprocess.run([QUERY] + args, capture_output=True, text=True, preexec_fn=limited)
    rc_o, body_o, _, _ = _cli(ORACLE_CLI, args)
    assert r.returncode == rc_o == 0, r.stderr[-2000:]
    body_p = [l for l in r.stdout.splitlines() if not l.startswith("Searched ")]
    assert sorted(body_p) == sorted(body_o)
    assert body_p[0] == f"Searching {n_files} files..."


def test_package_before_torch_shares_one_hip_runtime(tmp_path):
    """PyTorch ships its own libamdhip64 (same SONAME as ROCm's).  Whichever is imported first, the process must end up
    with ONE runtime: the package maps torch's copy before libpcq.so when torch is installed (binding._one_hip_runtime)."""
    code = r'''
import importlib, os, sys
sys.path.insert(0, %r)
pkg = importlib.import_module("adhoc-queries-pointclouds_amd")   # before torch
ctx = pkg.Context(0)
info = ctx.device_info()
import torch
assert torch.cuda.is_available(), "torch lost the device"
t = torch.arange(1000, device="cuda").sum().item()
assert t == 499500
d = ctx.alloc(64); ctx.memset(d, 0, 64); ctx.synchronize(); ctx.free(d)
ctx.close()
paths = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
print("RUNTIMES", len(paths), info["gcn_arch"])
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "RUNTIMES 1 gfx950" in r.stdout


def test_resident_dataset_batched_counts_equal_per_file_searches(oracle, q, files):
    """host/resident.cpp: the LAST files of a dataset loaded into HBM once; every count query is the per-file host
    prologue (early-out, box conversion) plus ONE batched launch.  Same totals as the per-file searches of the oracle."""
    lasts = [f for f in files if f.endswith(".last")]
    lib = q.lib
    lib.pcq_query_resident_load.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(C.c_void_p)]
    lib.pcq_query_resident_free.argtypes = [C.c_void_p]
    lib.pcq_query_resident_count_bounds.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                                                    C.POINTER(C.c_uint64)]
    lib.pcq_query_resident_count_class.argtypes = [C.c_void_p, C.c_uint8, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    arr = (C.c_char_p * len(lasts))(*[p.encode() for p in lasts])
    h = C.c_void_p()
    assert lib.pcq_query_resident_load(0, arr, len(lasts), C.byref(h)) == 0, lib.pcq_query_last_error()
    try:
        for bmin, bmax in BOXES:
            want = scanned_want = 0
            for path in lasts:
                oc = oracle.count_collector()
                assert oracle.search_file(path, 0, bmin, bmax, 0, oc)[0] == 0
                want += oc.point_count()
                oc.free()
            got, scanned = C.c_uint64(), C.c_uint64()
            for _ in range(2):  # the second time the segment table is already in HBM
                assert lib.pcq_query_resident_count_bounds(h, q.d3(bmin), q.d3(bmax), C.byref(got), C.byref(scanned)) == 0
                assert got.value == want, (bmin, bmax)
        for cls in (6, 134, 2, 19):
            want = 0
            for path in lasts:
                oc = oracle.count_collector()
                assert oracle.search_file(path, 1, None, None, cls, oc)[0] == 0
                want += oc.point_count()
                oc.free()
            got, scanned = C.c_uint64(), C.c_uint64()
            assert lib.pcq_query_resident_count_class(h, cls, C.byref(got), C.byref(scanned)) == 0
            assert got.value == want and scanned.value == sum(70_001 + 13 * f for f in (0, 1, 2, 3))
        # a box whose min exceeds its max panics like AABB::from_min_max (main.rs:80-91)
        assert lib.pcq_query_resident_count_bounds(h, q.d3((1, 1, 1)), q.d3((0, 2, 2)), C.byref(got), C.byref(scanned)) == -7
        # a .las file is refused: the batched kernels read LAST column blocks
        h2 = C.c_void_p()
        bad = (C.c_char_p * 1)([f for f in files if f.endswith(".las")][0].encode())
        assert lib.pcq_query_resident_load(0, bad, 1, C.byref(h2)) == -4
    finally:
        lib.pcq_query_resident_free(h)
