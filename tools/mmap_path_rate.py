"""Experiment: pcq_scan_host from a FRESH read-only file mapping (remapped every run, so every page faults
anew) vs pcq_scan_fd (pread), by number of copy threads.  (MADV_POPULATE_READ on the slices was tried too: no
difference.)"""
import importlib, mmap, os, sys, tempfile, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
binding = importlib.import_module("adhoc-queries-pointclouds_amd.binding")
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
n = 100_000_000
spec = specs.synth_ca13(points_per_file=n, files=1)[0]
with pkg.Context(0) as ctx, tempfile.TemporaryDirectory(dir="/tmp") as d:
    dptr = ctx.alloc(12 * n)
    ctx.synth_fill(spec, 0, n, dptr, None)
    host = np.empty(3 * n, dtype=np.int32)
    ctx.to_host(host, dptr)
    ctx.free(dptr)
    path = os.path.join(d, "positions.bin")
    with open(path, "wb") as f:
        f.write(b"\0" * 4096)
        host.tofile(f)
    del host
    fd = os.open(path, os.O_RDONLY)
    bmin, bmax = specs.box("ca13_XL")
    lmin, lmax = pkg.box_to_local(bmin, bmax, list(spec.scale), list(spec.offset))
    pred = pkg.Predicate.bounds(lmin, lmax)
    for threads in (1, 2, 4, 8, 16):
        ctx.set_option("copy_threads", threads)
        res = {}
        for mode in ("pread", "mmap"):
            times = []
            for _ in range(4):
                cc = ctx.count_collector()
                t0 = time.perf_counter()
                if mode == "pread":
                    cols = binding.make_columns(xyz=4096, n=n, scale=list(spec.scale), offset=list(spec.offset))
                    ctx.scan_fd(fd, cols, pred, cc)
                else:
                    m = mmap.mmap(fd, 0, prot=mmap.PROT_READ)
                    buf = np.frombuffer(m, dtype=np.uint8)
                    cols = binding.make_columns(xyz=buf.ctypes.data + 4096, n=n, scale=list(spec.scale), offset=list(spec.offset))
                    ctx.scan_host(cols, pred, cc)
                    del buf
                    m.close()
                cnt = cc.point_count()
                times.append(time.perf_counter() - t0)
                cc.free()
                assert cnt == n
            res[mode] = 12 * n / sorted(times[1:])[1] / 1e9
        print(f"copy threads {threads:2d}: pread {res['pread']:5.1f} GB/s   fresh mmap + memcpy {res['mmap']:5.1f} GB/s", flush=True)
    os.close(fd)
