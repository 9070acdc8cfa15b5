"""Do the scan kernels read PINNED HOST memory at the PCIe rate?  One 20 M-point column set in pinned memory (torch), scanned in place
by pcq_scan_dev (count and grid collectors) against the usual pcq_scan_host of the same bytes.  usage: zero_copy_probe.py [points]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
binding = importlib.import_module("adhoc-queries-pointclouds_amd.binding")
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
with pkg.Context(0) as ctx:
    spec = specs.synth_ca13(points_per_file=n)[5]
    xyz, cls = ctx.alloc(12 * n), ctx.alloc(n)
    ctx.synth_fill(spec, 0, n, xyz, cls)
    ctx.synchronize()
    hx = torch.empty(12 * n, dtype=torch.uint8).pin_memory()
    hc = torch.empty(n, dtype=torch.uint8).pin_memory()
    ctx.to_host(hx.numpy(), xyz)
    ctx.to_host(hc.numpy(), cls)
    bmin, bmax = specs.box("ca13_XL")
    lmin, lmax = pkg.box_to_local(bmin, bmax, list(spec.scale), list(spec.offset))
    pred = pkg.Predicate.bounds(lmin, lmax)
    for name, mk in (("count", lambda: ctx.count_collector()), ("grid 100 m", lambda: ctx.grid_collector(bmin, bmax, 100.0))):
        for how in ("resident", "pinned in place", "scan_host"):
            if how == "resident":
                cols = binding.make_columns(xyz=xyz, cls=cls, n=n, scale=list(spec.scale), offset=list(spec.offset))
            else:
                cols = binding.make_columns(xyz=hx.data_ptr(), cls=hc.data_ptr(), n=n, scale=list(spec.scale), offset=list(spec.offset))
            best = None
            for rep in range(4):
                c = mk()
                ctx.synchronize()
                t0 = time.perf_counter()
                (ctx.scan_host if how == "scan_host" else ctx.scan_dev)(cols, pred, c)
                ctx.synchronize()
                dt = time.perf_counter() - t0
                k = c.point_count()
                c.free()
                best = dt if best is None or dt < best else best
            print("%-10s %-16s %8.3f ms  %6.1f GB/s of 13 B/point   result %d" % (name, how, best * 1e3, 13 * n / best / 1e9, k), flush=True)
