"""Measured read-only HBM streaming ceiling on this device, next to K1 (developer tool).
Rotates over 8 x 2 GB buffers (no Infinity-Cache re-reads), interleaved rounds, HIP events."""
import os
os.environ.setdefault("PCQ_LAB", "1")  # the kernel shapes / microbenchmarks swept here live in libpcq_lab.so (make -C csrc lab)
import ctypes as C, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
binding = importlib.import_module("adhoc-queries-pointclouds_amd.binding")
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")

n = 163_000_000
nfiles = 8
dev = torch.device("cuda:0")
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); stream = ts.cuda_stream
with pkg.Context(0) as ctx:
    ss = specs.synth_ca13(points_per_file=n, files=nfiles)
    bufs, cols, preds = [], [], []
    bmin, bmax = specs.box("ca13_XL")
    for s in ss:
        x = torch.empty(n * 12, dtype=torch.uint8, device=dev)
        ctx.synth_fill(s, 0, n, x.data_ptr(), None, stream)
        bufs.append(x)
        cols.append(binding.make_columns(xyz=x.data_ptr(), n=n, scale=list(s.scale), offset=list(s.offset)))
        lmin, lmax = pkg.box_to_local(bmin, bmax, list(s.scale), list(s.offset))
        preds.append(pkg.Predicate.bounds(lmin, lmax))
    torch.cuda.synchronize()
    counter = torch.zeros(2, dtype=torch.int64, device=dev)
    cc = ctx.count_collector(device_counter=counter.data_ptr())
    configs = [("K1", 0, 1, b) for b in (2, 3, 4)]
    for shape in (0, 1, 2, 3):
        for nt in (1, 0):
            for b in (2, 3, 4, 8):
                configs.append(("read", shape, nt, b))
    times = {c: [] for c in configs}
    k = 0
    for r in range(10):
        for cfg in configs:
            kind, shape, nt, b = cfg
            f = k % nfiles; k += 1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if kind == "K1":
                ctx.set_option("blocks_per_cu", b)
                e0.record(); ctx.scan_dev(cols[f], preds[f], cc, stream); e1.record()
            else:
                e0.record()
                rc = ctx.lib.pcq_membench_read(ctx.handle, C.c_void_p(bufs[f].data_ptr()), n * 12, shape, nt, b, C.c_void_p(stream))
                assert rc == 0
                e1.record()
            e1.synchronize()
            if r >= 2: times[cfg].append(e0.elapsed_time(e1))
    rows = []
    for cfg, t in times.items():
        t.sort(); med = t[len(t) // 2]
        rows.append((n * 12 / med / 1e6, cfg))
    for gbs, cfg in sorted(rows, reverse=True):
        print(f"{gbs:8.1f} GB/s  {cfg[0]:5} shape={cfg[1]} nt={cfg[2]} blocks/cu={cfg[3]}", flush=True)
    cc.free()
