import importlib, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
binding = importlib.import_module("adhoc-queries-pointclouds_amd.binding")
n = 163_000_000
cls = np.random.default_rng(1).choice(np.array([1, 2, 2, 5, 6], dtype=np.uint8), n)
with pkg.Context(0) as ctx:
    cols = binding.make_columns(cls=cls.ctypes.data, n=n, scale=[0.01] * 3, offset=[0.0] * 3)
    pred = pkg.Predicate.classification(6)
    for chunk in (2 << 20, 2 << 20):
        ctx.set_option("chunk_points", chunk)
        ts = []
        for _ in range(5):
            cc = ctx.count_collector(); t = time.perf_counter(); ctx.scan_host(cols, pred, cc); c = cc.point_count(); ts.append(time.perf_counter() - t); cc.free()
        print(f"class count from host memory, {n} points: {sorted(ts)[2] * 1e3:.2f} ms = {n / sorted(ts)[2] / 1e9:.1f} GB/s, count {c} == {int((cls == 6).sum())}")
