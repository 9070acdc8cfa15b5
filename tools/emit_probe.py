"""Developer probe: buffer-collector scans of one resident file (for rocprofv3).  usage: emit_probe.py QUERY [POINTS] [REPEATS]
FRAC=<0..1>: the query box is the file's own box cut to that fraction of its x range (instead of QUERY's box);
SORTED=1: the file's points are sorted along x first (matches of such a box are then one contiguous run of the file)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
binding = importlib.import_module("adhoc-queries-pointclouds_amd.binding")
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
q, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 163_000_000
with pkg.Context(0) as ctx:
    if os.environ.get("EMIT_SPARSE_MAX") is not None:  # threshold of the sparse writer (0 = dense writer only)
        ctx.set_option("emit_sparse_max", int(os.environ["EMIT_SPARSE_MAX"]))
    if os.environ.get("EMIT_PARK_MAX") is not None:
        ctx.set_option("emit_park_max", int(os.environ["EMIT_PARK_MAX"]))
    spec = specs.synth_ca13(points_per_file=n)[5]
    xyz, cls = ctx.alloc(12 * n), ctx.alloc(n)
    ctx.synth_fill(spec, 0, n, xyz, cls)
    ctx.synchronize()
    if os.environ.get("SORTED"):
        import numpy as np
        host = np.zeros((n, 3), dtype=np.int32)
        ctx.to_host(host, xyz)
        host = host[np.argsort(host[:, 0], kind="stable")]
        ctx.to_device(xyz, np.ascontiguousarray(host))
        del host
    cols = binding.make_columns(xyz=xyz, cls=cls, n=n, scale=list(spec.scale), offset=list(spec.offset))
    bmin, bmax = specs.box(q)
    if os.environ.get("FRAC"):
        hf = specs.header_fields(spec)
        lo, hi = list(hf["min"]), list(hf["max"])
        hi[0] = lo[0] + float(os.environ["FRAC"]) * (hi[0] - lo[0])
        bmin, bmax, q = lo, hi, "x<%s" % os.environ["FRAC"]
    lmin, lmax = pkg.box_to_local(bmin, bmax, list(spec.scale), list(spec.offset))
    b = ctx.buffer_collector()
    for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
        b.reset()
        t0 = time.perf_counter()
        ctx.scan_dev(cols, pkg.Predicate.bounds(lmin, lmax), b)
        ctx.synchronize()
        t1 = time.perf_counter()
        k = b.point_count()
        print(q, "matches", k, "scan %.2f ms  %.0f GB/s of (12 + 1 B read per point, 31 B written per match)" % ((t1 - t0) * 1e3, (13 * n + 31 * k) / (t1 - t0) / 1e9), flush=True)
    b.free()
