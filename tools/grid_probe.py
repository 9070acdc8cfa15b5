"""Developer probe: grid-collector scans of one resident file (for rocprofv3 --kernel-trace).
usage: grid_probe.py QUERY CELL [POINTS] [REPEATS]; GRID_F2=<n> forces the second-level fan-out;
GRID_AGG=<0|1|2>[,...] sets pass 0's tile fold (0 = while it pays, 1 = every tile, 2 = never), one value per repeat in turn;
COHERENT=<metres> reorders the file into x/y strips of that width (points sorted along each strip, like scan
lines) instead of the generator's random order."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
binding = importlib.import_module("adhoc-queries-pointclouds_amd.binding")
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
q, cell, n = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 163_000_000
if os.environ.get("COHERENT"):
    import torch  # plumbing for the device sort; initialised before the context
    torch.cuda.init()
with pkg.Context(0) as ctx:
    spec = specs.synth_ca13(points_per_file=n)[5]
    xyz, cls = ctx.alloc(12 * n), ctx.alloc(n)
    ctx.synth_fill(spec, 0, n, xyz, cls)
    ctx.synchronize()
    if os.environ.get("COHERENT"):
        import numpy as np
        import torch
        width = int(float(os.environ["COHERENT"]) / spec.scale[1])
        host = np.zeros((n, 3), dtype=np.int32)
        ctx.to_host(host, xyz)
        t = torch.from_numpy(host).cuda()
        zwidth = int(float(os.environ["COHERENT"]) / spec.scale[2])
        key = ((t[:, 1].long() // width) * 4096 + (t[:, 2].long() // zwidth + 2048)) * (1 << 32) + (t[:, 0].long() + (1 << 31))
        t = t[torch.argsort(key)].contiguous().cpu().numpy()
        ctx.to_device(xyz, t)
        del key
    cols = binding.make_columns(xyz=xyz, cls=cls, n=n, scale=list(spec.scale), offset=list(spec.offset))
    bmin, bmax = specs.box(q)
    lmin, lmax = pkg.box_to_local(bmin, bmax, list(spec.scale), list(spec.offset))
    ctx.set_option("grid_f2", int(os.environ.get("GRID_F2", "0")))
    for opt_name in ("grid_block_pad", "grid_stream", "grid_tuple16"):  # GRID_BLOCK_PAD / GRID_STREAM / GRID_TUPLE16
        if os.environ.get(opt_name.upper()) is not None:
            ctx.set_option(opt_name, int(os.environ[opt_name.upper()]))
    aggs = [int(v) for v in os.environ.get("GRID_AGG", "").split(",") if v]  # several: one after the other, same buffers
    for it in range(int(sys.argv[4]) if len(sys.argv) > 4 else 2):
        if aggs:
            ctx.set_option("grid_agg", aggs[it % len(aggs)])
        t0 = time.perf_counter()
        g = ctx.grid_collector(bmin, bmax, cell)
        t1 = time.perf_counter()
        ctx.scan_dev(cols, pkg.Predicate.bounds(lmin, lmax), g)
        ctx.synchronize()
        t2 = time.perf_counter()
        k = g.point_count()
        t3 = time.perf_counter()
        g.free()
        t4 = time.perf_counter()
        print(q, cell, "cells", k, "new %.1f scan %.1f count %.1f free %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3),
              "f2", ctx.get_option("grid_last_f2"), "refolds", ctx.get_option("grid_refolds"), "agg", ctx.get_option("grid_agg"),
              "tuples", ctx.get_option("grid_last_tuples"), flush=True)
