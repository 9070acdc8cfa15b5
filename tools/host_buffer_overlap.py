"""Developer probe: pcq_scan_host of a file in host memory into a buffer collector and into a grid collector — for
`rocprofv3 --kernel-trace --memory-copy-trace`: do the H2D copies of chunk k+1 overlap the kernels of chunk k?
usage: host_buffer_overlap.py [POINTS]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
binding = importlib.import_module("adhoc-queries-pointclouds_amd.binding")
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
import _oracle  # only to generate the file image
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40_000_000
spec = specs.synth_ca13(points_per_file=n)[5]
image = _oracle.Oracle().synth_image(spec, transposed=True, threads=16)
otp = 227
base = image.ctypes.data + otp
cols = binding.make_columns(xyz=base, cls=base + 15 * n, n=n, scale=list(spec.scale), offset=list(spec.offset))
bmin, bmax = specs.box("ca13_XL")
lmin, lmax = pkg.box_to_local(bmin, bmax, list(spec.scale), list(spec.offset))
with pkg.Context(0) as ctx:
    for name, make in (("buffer", lambda: ctx.buffer_collector()), ("grid100", lambda: ctx.grid_collector(bmin, bmax, 100.0))):
        for rep in range(2):
            c = make()
            t0 = time.perf_counter()
            ctx.scan_host(cols, pkg.Predicate.bounds(lmin, lmax), c)
            t1 = time.perf_counter()
            k = c.point_count()
            t2 = time.perf_counter()
            print(f"{name}: {n} points from host memory in {(t1 - t0) * 1e3:.1f} ms = {13 * n / (t1 - t0) / 1e9:.1f} GB/s of column bytes; result {k} after {(t2 - t1) * 1e3:.1f} ms more", flush=True)
            c.free()
