"""Developer tool: the host layer's resident dataset (host/resident.cpp) on synthetic doc — 8 LAST files x 106.75 M points
written to /dev/shm, loaded into HBM once, then `--class 6` and `--bounds doc_L` count queries as ONE batched launch each.
Prints the wall time per query (launch + 8-byte read-back included) and the byte rate; run under
`rocprofv3 --kernel-trace --stats` for the kernel-only rate (k_class_count_batch_pipe<4>: 1 B/point, k_bounds_count_batch_pipe<2>:
12 B/point)."""
import ctypes as C
import importlib
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
import _oracle  # only to WRITE the synthetic files (the generator lives in the oracle)

points = int(sys.argv[1]) if len(sys.argv) > 1 else 106_750_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
d = tempfile.mkdtemp(prefix="pcq_resident_", dir="/dev/shm")
try:
    oracle = _oracle.Oracle()
    paths = []
    for i, s in enumerate(specs.synth_doc(points_per_file=points)):
        p = os.path.join(d, f"doc{i}.last")
        oracle.synth_write(s, p)
        paths.append(p)
    pkg.load_library()
    lib = C.CDLL(os.path.join(ROOT, "adhoc-queries-pointclouds_amd", "libpcq_query.so"))
    lib.pcq_query_resident_load.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(C.c_void_p)]
    lib.pcq_query_resident_count_bounds.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.pcq_query_resident_count_class.argtypes = [C.c_void_p, C.c_uint8, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.pcq_query_resident_free.argtypes = [C.c_void_p]
    lib.pcq_query_last_error.restype = C.c_char_p
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    h = C.c_void_p()
    t0 = time.perf_counter()
    rc = lib.pcq_query_resident_load(0, arr, len(paths), C.byref(h))
    assert rc == 0, lib.pcq_query_last_error()
    print(f"loaded {len(paths)} files x {points} points in {time.perf_counter() - t0:.2f} s", flush=True)
    got, scanned = C.c_uint64(), C.c_uint64()
    for name, call, bytes_per_point in (
            ("class 6", lambda: lib.pcq_query_resident_count_class(h, 6, C.byref(got), C.byref(scanned)), 1),
            ("bounds doc_L", lambda: lib.pcq_query_resident_count_bounds(h, (C.c_double * 3)(*specs.box("doc_L")[0]), (C.c_double * 3)(*specs.box("doc_L")[1]),
                                                                        C.byref(got), C.byref(scanned)), 12)):
        assert call() == 0
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            assert call() == 0
            ts.append(time.perf_counter() - t0)
        ts.sort()
        med = ts[len(ts) // 2]
        print(f"{name}: {got.value} matches, {scanned.value} points scanned, {med * 1e3:.3f} ms per query (median of {reps}), "
              f"{scanned.value * bytes_per_point / med / 1e9:.0f} GB/s incl. launch and read-back", flush=True)
    lib.pcq_query_resident_free(h)
finally:
    shutil.rmtree(d, ignore_errors=True)
