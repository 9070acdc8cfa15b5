"""CPU-only checks of the product side: the C-ABI libraries load and export every symbol the headers
declare, the host logic (header parse, box conversion, argument handling) matches the oracle and the
golden vectors, and — without a GPU — the product fails loudly instead of falling back to a CPU path.
No compute call is made here.
"""
import ctypes as C
import importlib
import json
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "adhoc-queries-pointclouds_amd")
G = json.load(open(os.path.join(HERE, "golden", "expected.json")))

pkg = importlib.import_module("adhoc-queries-pointclouds_amd")


def fhex(s):
    return float("nan") if s == "nan" else float.fromhex(s)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_libpcq_exports_every_declared_symbol():
    declared = pkg.declared_symbols(["pcq.h", "pcq_synth.h"])
    assert len(declared) >= 29
    exported = pkg.exported_symbols(pkg.lib_path())
    assert [s for s in declared if s not in exported] == []
    lib = pkg.load_library()  # dlopen resolves libamdhip64 etc.
    assert lib.pcq_abi_version() == 6


def test_libpcq_query_exports_every_declared_symbol():
    declared = pkg.declared_symbols(["pcq_query.h"])
    declared = [s for s in declared if s.startswith("pcq_query")]
    assert len(declared) >= 15
    path = os.path.join(PKG, "libpcq_query.so")
    exported = pkg.exported_symbols(path)
    assert [s for s in declared if s not in exported] == []
    C.CDLL(path)


def test_product_library_ships_no_lab_code():
    """libpcq.so = the shipped kernels; the superseded kernel shapes, the HBM read microbenchmarks and the device LZ4
    inflater measured in round 1 are not in it (csrc/lab/ -> libpcq_lab.so, loaded only by tools/ with PCQ_LAB=1)."""
    syms = subprocess.run(["nm", "-D", "--defined-only", pkg.lib_path()], capture_output=True, text=True).stdout
    assert "pcq_scan_dev" in syms
    for name in ("membench", "lz4_inflate"):
        assert name not in syms, name
    all_syms = subprocess.run(["nm", "-C", pkg.lib_path()], capture_output=True, text=True).stdout
    for kernel in ("k_bounds_count_xyz12", "k_bounds_count_w1<", "k_bounds_count_batch_w1", "k_class_count_batch_w1", "k_class_count_u8"):
        assert kernel not in all_syms, kernel
    for kernel in ("k_bounds_count_w1_pipe<2>", "k_bounds_count_batch_pipe<2>", "k_class_count_pipe<4>", "k_class_count_batch_pipe<4>"):
        assert kernel in all_syms, kernel
    ctx_opts = open(os.path.join(ROOT, "include", "pcq.h")).read()
    assert '"k1_variant" (bounds-count kernel variant' not in ctx_opts


def test_no_oracle_in_the_product_binaries():
    """The product must not link, load or embed the oracle."""
    for name in ("libpcq.so", "libpcq_query.so", os.path.join("host", "query")):
        path = os.path.join(PKG, name)
        needed = subprocess.run(["readelf", "-d", path], capture_output=True, text=True).stdout
        assert "liboracle" not in needed
        syms = subprocess.run(["nm", "-D", path], capture_output=True, text=True).stdout
        assert "pcqo_" not in syms
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".cpp", ".hip", ".h", ".hpp", ".py")):
                text = open(os.path.join(dirpath, f)).read()
                assert "pcq_oracle.h" not in text and "liboracle" not in text, f


def test_the_query_binary_has_no_test_hooks():
    """The parallel driver's test hooks (device slots with repeats, an injected all-reduce failure) are RunOptions fields that
    only the test entry of the C view sets (pcq_query_main_with_hooks); no environment variable reaches them, so the shipped
    binary cannot be told that it has two GPUs."""
    for name in (os.path.join("host", "query"), "libpcq_query.so", "libpcq.so"):
        out = subprocess.run(["strings", os.path.join(PKG, name)], capture_output=True, text=True).stdout
        assert "PCQ_TEST_" not in out, name
    for f in ("run_search.cpp", "core.cpp", "search.cpp", "main.cpp"):
        assert "PCQ_TEST_" not in open(os.path.join(PKG, "host", f)).read(), f


def _schedule(qlib, cost, ready_ms, ms_per_unit=1.0):
    n, k = len(cost), len(ready_ms)
    slot, home, end = (C.c_int * n)(), (C.c_int * n)(), C.c_double()
    qlib.pcq_query_simulate_schedule.argtypes = [C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_double), C.c_int, C.c_double,
                                                 C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    assert qlib.pcq_query_simulate_schedule((C.c_uint64 * n)(*cost), n, (C.c_double * k)(*ready_ms), k, ms_per_unit, slot, home, C.byref(end)) == 0
    return list(slot), list(home), end.value


def test_file_to_device_assignment_with_staggered_context_readiness(qlib):
    """run_search_parallel's file -> device-slot schedule (main.rs:153-161: whichever rayon thread is free takes the next file),
    simulated without a GPU.  Every slot starts with its own longest-processing-time share (equal files: file i -> slot i % N,
    the rule sharding.py applies across processes); a slot that is ready late keeps its files unless somebody would idle;
    a slot that runs dry takes the smallest file of the slot with the most work left."""
    sharding = importlib.import_module("adhoc-queries-pointclouds_amd.sharding")
    # (1) 16 equal files, 8 slots, all contexts ready together: nobody steals, file i is scanned by slot i % 8
    slot, home, end = _schedule(qlib, [20] * 16, [50.0] * 8)
    assert slot == home == [i % 8 for i in range(16)] and end == 50.0 + 40.0
    for r in range(8):
        assert sharding.assign_files(16, 8, r, points=[20] * 16) == [i for i in range(16) if slot[i] == r]
    # (2) the contexts come up 50 ms apart (what ONE process-wide start-up mutex gave: 8 x 50-230 ms): the early slots take
    # over files of the late ones instead of idling, every file is scanned exactly once, and the last worker ends well before
    # the last context + its own share would
    slot, home, end = _schedule(qlib, [20] * 16, [50.0 * (k + 1) for k in range(8)])
    assert sorted(set(slot)) != [0] and all(0 <= v < 8 for v in slot) and len(slot) == 16
    assert home == [i % 8 for i in range(16)]
    assert slot.count(0) > 2 and slot[0] == 0 and slot[8] == 0      # slot 0: its own two first, then others'
    assert end < 400.0 + 40.0                                       # (slot 7 alone: ready at 400 ms + its two files)
    # (3) start-up in parallel (one mutex per device): ready within 10 ms of each other -> the LPT shares hold
    slot, home, end = _schedule(qlib, [20] * 16, [50.0 + k for k in range(8)])
    assert slot == home
    # (4) unequal files: longest first; the share of every slot within one largest file of the mean
    cost = [100, 90, 80, 70, 60, 50, 40, 30, 20, 10, 5, 5, 5, 5]
    slot, home, end = _schedule(qlib, cost, [0.0, 0.0, 0.0])
    loads = [sum(c for c, h in zip(cost, home) if h == k) for k in range(3)]
    assert max(loads) - min(loads) <= 100 and sorted(slot) == sorted(home)
    assert end <= sum(cost) / 3 + 100
    # (5) one slot: everything in LPT order on it; no file lost with more slots than files
    slot, home, end = _schedule(qlib, [3, 1, 2], [0.0])
    assert slot == [0, 0, 0] and end == 6.0
    slot, home, end = _schedule(qlib, [7, 9], [0.0, 0.0, 0.0, 0.0])
    assert sorted(slot) == [0, 1]


def test_box_to_local_matches_golden_and_oracle(oracle):
    for c in G["box_to_local"]:
        args = ([fhex(v) for v in c["bmin"]], [fhex(v) for v in c["bmax"]], [fhex(v) for v in c["scale"]],
                [fhex(v) for v in c["offset"]])
        if c["panic"]:
            with pytest.raises(pkg.PcqError) as e:
                pkg.box_to_local(*args)
            assert e.value.code == -7  # PCQ_ERR_PANIC
        else:
            assert pkg.box_to_local(*args) == (c["lmin"], c["lmax"])
    rng = np.random.default_rng(7)
    for _ in range(2000):
        lo = rng.uniform(-1e6, 1e6, 3)
        hi = lo + rng.uniform(0, 1e5, 3)
        sc = 10.0 ** rng.integers(-4, 1, 3)
        off = rng.uniform(-1e5, 1e5, 3)
        try:
            want = oracle.box_to_local(lo, hi, sc, off)
        except Exception:
            with pytest.raises(pkg.PcqError):
                pkg.box_to_local(lo, hi, sc, off)
            continue
        assert pkg.box_to_local(lo, hi, sc, off) == want


class HeaderInfo(C.Structure):
    _fields_ = [("version_major", C.c_uint8), ("version_minor", C.c_uint8), ("point_data_record_format", C.c_uint8),
                ("_pad", C.c_uint8), ("header_size", C.c_uint16), ("point_data_record_length", C.c_uint16),
                ("offset_to_point_data", C.c_uint32), ("_pad2", C.c_uint32), ("number_of_points", C.c_uint64),
                ("scale", C.c_double * 3), ("offset", C.c_double * 3), ("min", C.c_double * 3), ("max", C.c_double * 3)]


@pytest.fixture(scope="module")
def qlib():
    lib = C.CDLL(os.path.join(PKG, "libpcq_query.so"))
    lib.pcq_query_last_error.restype = C.c_char_p
    lib.pcq_query_parse_las_header.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(HeaderInfo)]
    lib.pcq_query_parse_aabb.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.pcq_query_is_valid_file.argtypes = [C.c_char_p]
    lib.pcq_query_collector_new_count.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    return lib


def _mutations(image: np.ndarray):
    yield "ok", image
    yield "short", image[:100]
    yield "sig", np.concatenate([np.frombuffer(b"XASF", dtype=np.uint8), image[4:]])
    for fmt in (0x82, 11, 15, 6, 0x12):
        m = image.copy()
        m[104] = fmt
        yield f"fmt{fmt}", m
    m = image.copy()
    m[105:107] = np.frombuffer(np.uint16(10).tobytes(), dtype=np.uint8)  # record length < format length
    yield "reclen", m
    m = image.copy()
    m[25] = 4  # LAS 1.4 header is longer than the 227 bytes + data available? (tail fields read)
    yield "v14", m
    m = image.copy()
    m[25] = 4
    m[104] = 6
    m[105:107] = np.frombuffer(np.uint16(30).tobytes(), dtype=np.uint8)
    m[107:111] = 0  # legacy count 0 -> large count from the 1.4 tail
    yield "v14_fmt6", m


def test_host_header_parse_matches_oracle(oracle, qlib):
    import _oracle
    last = np.fromfile(os.path.join(HERE, "golden", "tiny_fmt2.last"), dtype=np.uint8)
    for name, img in _mutations(last):
        img = np.ascontiguousarray(img)
        for mask in (0, 1):
            h = HeaderInfo()
            rc = qlib.pcq_query_parse_las_header(img.ctypes.data_as(C.c_void_p), img.size, mask, C.byref(h))
            try:
                oh = oracle.parse_header(img.tobytes(), bool(mask))
                orc = 0
            except _oracle.OracleError as e:
                orc, oh = e.code, None
            assert rc == orc, (name, mask, qlib.pcq_query_last_error())
            if rc == 0:
                assert (h.number_of_points, h.point_data_record_format, h.point_data_record_length, h.offset_to_point_data) == \
                       (oh.number_of_points, oh.point_data_record_format, oh.point_data_record_length, oh.offset_to_point_data), name
                assert list(h.scale) == list(oh.scale) and list(h.offset) == list(oh.offset)
                assert list(h.min) == list(oh.min) and list(h.max) == list(oh.max)


def test_parse_aabb_and_is_valid_file(qlib):
    mn, mx = (C.c_double * 3)(), (C.c_double * 3)()
    assert qlib.pcq_query_parse_aabb(b"665000;3910000;0;705000;3950000;480", mn, mx) == 0
    assert list(mn) == [665000.0, 3910000.0, 0.0] and list(mx) == [705000.0, 3950000.0, 480.0]
    assert qlib.pcq_query_parse_aabb(b"-23.108;-21.261;-10.029;28.588;27.123;5.959", mn, mx) == 0
    assert list(mn) == [-23.108, -21.261, -10.029]
    assert qlib.pcq_query_parse_aabb(b"1;2;3;4;5", mn, mx) == -8          # main.rs:61-63
    assert qlib.pcq_query_parse_aabb(b"1;2;3;4;5;x", mn, mx) == -8        # main.rs:65-78
    assert qlib.pcq_query_parse_aabb(b"1;2;3;4;5; 6", mn, mx) == -8       # Rust's parse rejects whitespace
    assert qlib.pcq_query_parse_aabb(b"5;0;0;1;1;1", mn, mx) == -7        # from_min_max panics (min > max)
    for name, ok in (("a.las", 1), ("a.laz", 1), ("b.last", 1), ("c.lazer", 1), ("d.LAS", 0), ("e.txt", 0), ("las", 0),
                     (".las", 0), ("dir.las/x", 0), ("x.laser", 0)):
        assert qlib.pcq_query_is_valid_file(name.encode()) == ok, name  # main.rs:185-189


@pytest.mark.skipif(_has_gpu(), reason="checks the behaviour on a machine WITHOUT a GPU")
def test_product_fails_loudly_without_a_gpu(qlib):
    with pytest.raises(pkg.PcqError) as e:
        pkg.Context(0)
    assert e.value.code == -9 and "no CPU path" in e.value.message  # PCQ_ERR_HIP
    h = C.c_void_p()
    assert qlib.pcq_query_collector_new_count(0, C.byref(h)) == -9


def _run(exe, args):
    r = subprocess.run([exe] + args, capture_output=True, text=True)
    return r.returncode, r.stdout, r.stderr


@pytest.mark.parametrize("args", [
    [],                                                                    # missing --input
    ["-i", "/nonexistent/path", "--bounds", "0;0;0;1;1;1"],                # main.rs:30-35
    ["-i", "DIR", "--bounds", "0;0;0;1;1;1", "--class", "6"],              # main.rs:238-240
    ["-i", "DIR"],                                                         # main.rs:242-244
    ["-i", "DIR", "--bounds", "nonsense"],                                 # expect() panic, exit 101
    ["-i", "DIR", "--bounds", "2;0;0;1;1;1"],                              # from_min_max panic
    ["-i", "DIR", "--class", "256"],                                       # u8 parse panic
    ["-i", "DIR", "--class", "-1"],
    ["-i", "DIR", "--bounds", "0;0;0;1;1;1", "--density", "abc"],
    ["-i", "DIR", "--bounds", "0;0;0;1;1;1", "-o", "/nonexistent/out"],    # FileDumper::new
    ["-i", "DIR", "--frobnicate"],
])
def test_cli_argument_errors_match_oracle_cli(oracle, tmp_path, args):
    """Argument handling happens before any GPU work: same exit status as the oracle CLI."""
    d = tmp_path / "data"
    d.mkdir()
    args = [str(d) if a == "DIR" else a for a in args]
    rc_p, out_p, err_p = _run(os.path.join(PKG, "host", "query"), args)
    rc_o, out_o, err_o = _run(os.path.join(ROOT, "oracle", "query_oracle"), args)
    assert rc_p == rc_o and rc_p != 0, (args, rc_p, rc_o, err_p, err_o)
    assert out_p == out_o == ""


def test_get_total_bounds_is_the_union_of_header_boxes(oracle, qlib, tmp_path):
    """main.rs:94-120 — used as the grid box for `--class ... --density` (main.rs:255-259)."""
    specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
    paths, mins, maxs = [], [], []
    for i, s in enumerate(specs.synth_ca13(points_per_file=101, files=5) + specs.synth_doc(97)[:2]):
        p = str(tmp_path / (f"t{i}.last" if i % 2 else f"t{i}.las"))
        oracle.synth_write(s, p)
        h = oracle.parse_header(open(p, "rb").read(400))
        paths.append(p)
        mins.append(list(h.min))
        maxs.append(list(h.max))
    qlib.pcq_query_get_total_bounds.argtypes = [C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    mn, mx = (C.c_double * 3)(), (C.c_double * 3)()
    assert qlib.pcq_query_get_total_bounds(arr, len(paths), mn, mx) == 0
    assert list(mn) == [min(m[a] for m in mins) for a in range(3)]
    assert list(mx) == [max(m[a] for m in maxs) for a in range(3)]
    # a file that is not a LAS file fails the whole call
    bad = str(tmp_path / "bad.last")
    open(bad, "wb").write(b"not a las file at all")
    arr2 = (C.c_char_p * 1)(bad.encode())
    assert qlib.pcq_query_get_total_bounds(arr2, 1, mn, mx) == -2


def test_run_query_experiments_driver_protocol(oracle, tmp_path):
    """host/run_query_experiments mirrors query/src/bin/run_query_experiments.rs: directory layout
    <root>/<dataset>/<ext>/, 5 runs, `name;mean;median;stddev` lines.  Driven here with the oracle CLI as
    the `query` executable (no GPU)."""
    import importlib
    import subprocess
    specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
    root = tmp_path / "data"
    for name, ss in (("navvis3", specs.synth_navvis(points_per_file=3000)), ("doc", specs.synth_doc(points_per_file=1000, files=2)),
                     ("ca13", specs.synth_ca13(points_per_file=1000, files=3))):
        for ext in ("las", "last"):
            d = root / name / ext
            d.mkdir(parents=True)
            for i, s in enumerate(ss):
                oracle.synth_write(s, str(d / f"f{i}.{ext}"))
    drv = os.path.join(PKG, "host", "run_query_experiments")
    q = os.path.join(ROOT, "oracle", "query_oracle")

    def run(exp, *extra):
        settle = [] if "--settle-ms" in extra else ["--settle-ms", "0"]  # (default: 1 s in front of every run, for `sync; purge`)
        r = subprocess.run([drv, "-i", str(root), "-e", str(exp), "--extensions", "las,last", "--query", q, *settle, *extra],
                           capture_output=True, text=True)
        return r.returncode, r.stdout.splitlines(), r.stderr

    rc, lines, err = run(1, "--runs", "3")
    assert rc == 0, err
    want = [f"navvis3_{b}_{k}_{e}" for b in ("s", "l", "xl") for k in ("full", "lod") for e in ("las", "last")]
    assert [l.split(";")[0] for l in lines] == want  # run_query_experiments.rs:153-190, :257-285
    for l in lines:
        name, mean, med, sd = l.split(";")
        assert float(mean) > 0 and float(med) > 0 and float(sd) >= 0
    assert "Running experiments... Output is: experiment_name;mean;median;stddev with runtimes in seconds" in err
    assert "Experiment navvis3_s_las..." in err
    rc, lines, err = run(5, "--runs", "1")
    assert rc == 0 and [l.split(";")[0] for l in lines] == ["ca13_building_las", "ca13_building_last", "ca13_noclass_las", "ca13_noclass_last"]
    assert all(l.endswith(";0") for l in lines)  # one run: stddev 0
    rc, lines, err = run(4, "--runs", "1", "--cold")
    assert rc == 0 and len(lines) == 4
    import time
    t0 = time.perf_counter()
    rc, lines, err = run(5, "--runs", "2", "--settle-ms", "150")  # the pause stands outside the timed region
    assert rc == 0 and len(lines) == 4 and time.perf_counter() - t0 >= 8 * 0.15
    assert all(float(l.split(";")[1]) < 0.15 for l in lines)
    assert run(6)[0] == 1 and "Invalid experiment ID 6" in run(6)[2]
    # a format the query cannot search fails the run, like the reference's `?` on the child's exit status
    rc, lines, err = subprocess.run([drv, "-i", str(root), "-e", "1", "--query", q, "--runs", "1", "--settle-ms", "0"], capture_output=True, text=True).returncode, None, None
    assert rc == 1


def test_copy_pool_under_thread_sanitizer(tmp_path):
    """csrc/copy_pool.h hands a staging chunk to helper threads slice by slice.  Back-to-back jobs with different
    slice counts (positions 12 n, class n, colour 6 n) must never let a helper that is still leaving one job draw a
    ticket of the next: ThreadSanitizer build (CPU only), every copy compared with its source."""
    import shutil
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "copy_pool_tsan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-I" + os.path.join(PKG, "csrc"),
           os.path.join(ROOT, "tests", "native", "copy_pool_tsan_driver.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr and "cannot find" in r.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, "250", "7"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and r.stdout.startswith("ok 250"), (r.stdout[-500:], r.stderr[-3000:])


def test_every_entry_point_runs_on_the_context_device():
    """A driver thread that never chose a device sits on device 0; draining the collectors of GPU k from it must
    still allocate and launch on GPU k.  Every extern "C" function that takes a context, a collector or an index
    and touches the HIP runtime opens with the device guard (pcq_internal.h), or sets the device itself."""
    import re
    exempt = {"pcq_last_error", "pcq_abi_version", "pcq_init", "pcq_get_device_info", "pcq_ctx_stream", "pcq_get_option",
              "pcq_bind_thread_near_device", "pcq_box_to_local", "pcq_collector_has_points", "pcq_collector_grid_params",
              "pcq_collector_new_buffer",  # new_collector() sets the device
              "pcq_allreduce_sum_u64",     # sets each rank's device around its own calls
              "pcq_allreduce_prepare"}     # takes a device list, no context; restores the device it found
    checked = 0
    for f in sorted(os.listdir(os.path.join(PKG, "csrc"))):
        if not f.endswith(".hip"):
            continue
        text = open(os.path.join(PKG, "csrc", f)).read()
        for m in re.finditer(r'extern "C"[^;{]*?\b(pcq_\w+)\s*\(', text):
            name = m.group(1)
            if name in exempt:
                continue
            body_start = text.index("{", m.end())
            head = text[body_start:body_start + 400]
            assert "PCQ_ON_DEVICE_OF_" in head, f"{f}: {name} does not open with the device guard"
            checked += 1
    assert checked >= 25


def _kernel_asm(hip_file):
    """gfx950 assembly of one translation unit of the library (hipcc cross-compiles without a GPU)."""
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                          "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc"), "-S", "--cuda-device-only", "-o", "-",
                          os.path.join(PKG, "csrc", hip_file)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout.split("\n")


def _kernel_bodies(lines, name_part):
    for i, l in enumerate(lines):
        if l.startswith("_Z") and name_part in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]) and ":" in l:
            j = i
            while not lines[j].startswith(".Lfunc_end"):
                j += 1
            yield l.split(":")[0], lines[i:j]


def test_hot_loops_keep_their_loads_in_flight():
    """Two properties of the compiled gfx950 code that decide the grid collector's speed and that a source edit can lose
    without any test noticing (both were lost once in round 4): pass 0 must not wait for a class byte
    right behind its load — that wait also covers the next tile's positions, the kernel's prefetch — and the streaming fold's
    main loop (its longest stretch without a barrier) must hold no scratch access."""
    for name, body in _kernel_bodies(_kernel_asm("grid_pass0.hip"), "k_p0_part"):
        for n, l in enumerate(body):
            if "global_load_ubyte" in l or "global_load_ushort" in l:
                assert not any("vmcnt(0)" in x for x in body[n + 1:n + 3]), (name, n, l.strip())
    found = 0
    for name, body in _kernel_bodies(_kernel_asm("grid_fold_stream.hip"), "k_fold_stream"):
        if "ELb0ELb0E" not in name:  # the single-entry, 16-byte form: the one that runs on a file's own grid
            continue
        found += 1
        barriers = [n for n, l in enumerate(body) if l.strip().startswith("s_barrier")]
        assert len(barriers) >= 4, name
        a, b = max(zip(barriers, barriers[1:]), key=lambda ab: ab[1] - ab[0])
        loop = body[a:b]
        assert len(loop) > 1000  # (the stream is there: no barrier inside)
        assert not [l for l in loop if l.strip().startswith("scratch_")], name
    assert found == 1
