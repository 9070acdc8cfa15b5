"""PCIe-inclusive rate of the file path (pcq_scan_fd): positions block of a LAST-like file in the page cache
-> parallel pread into pinned staging -> hipMemcpyAsync -> K1, by copy threads and chunk size."""
import importlib, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
binding = importlib.import_module("adhoc-queries-pointclouds_amd.binding")
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
spec = specs.synth_ca13(points_per_file=n, files=1)[0]
with pkg.Context(0) as ctx, tempfile.TemporaryDirectory(dir="/tmp") as d:
    dptr = ctx.alloc(12 * n)
    ctx.synth_fill(spec, 0, n, dptr, None)
    host = np.empty(3 * n, dtype=np.int32)
    ctx.to_host(host, dptr)
    ctx.free(dptr)
    path = os.path.join(d, "positions.bin")
    with open(path, "wb") as f:
        f.write(b"\0" * 256)  # a header-sized prefix, so that the block does not start at offset 0
        host.tofile(f)
    del host
    fd = os.open(path, os.O_RDONLY)
    os.pread(fd, 1, 0)
    bmin, bmax = specs.box("ca13_XL")
    lmin, lmax = pkg.box_to_local(bmin, bmax, list(spec.scale), list(spec.offset))
    cols = binding.make_columns(xyz=256, n=n, scale=list(spec.scale), offset=list(spec.offset))
    pred = pkg.Predicate.bounds(lmin, lmax)
    out = {}
    print("GPU on NUMA node", ctx.get_option("numa_node"), flush=True)
    for numa, threads, chunk in [(nl, t, 2 << 20) for t in (1, 2, 4, 8) for nl in (0, 1, 0, 1)]:
        ctx.set_option("chunk_points", chunk)
        ctx.set_option("copy_threads", threads)
        ctx.set_option("numa_local", numa)
        times = []
        for _ in range(5):
            cc = ctx.count_collector()
            t0 = time.perf_counter()
            ctx.scan_fd(fd, cols, pred, cc)
            cnt = cc.point_count()
            times.append(time.perf_counter() - t0)
            cc.free()
        times = sorted(times[1:])
        med = times[len(times) // 2]
        assert cnt == n
        out.setdefault(f"threads_{threads}_numa_local_{numa}", []).append(12 * n / med / 1e9)
        print(f"copy threads {threads:2d}, staging + helpers on the GPU's node: {'yes' if numa else 'no '}  {12 * n / med / 1e9:6.1f} GB/s ({med * 1e3:.1f} ms)", flush=True)
    os.close(fd)
    print(json.dumps({"points": n, "path": "page cache -> parallel pread -> pinned staging -> hipMemcpyAsync -> K1", **out}))
