"""Developer tool: the software-pipelined per-file K1 (variant 12) by absolute grid size (workgroups of one wave)."""
import os
os.environ.setdefault("PCQ_LAB", "1")  # the kernel shapes / microbenchmarks swept here live in libpcq_lab.so (make -C csrc lab)
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
binding = importlib.import_module("adhoc-queries-pointclouds_amd.binding")
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
n, files = 163_000_000, 8
dev = torch.device("cuda:0")
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); stream = ts.cuda_stream
with pkg.Context(0) as ctx:
    cols, preds, keep = [], [], []
    bmin, bmax = specs.box("ca13_XL")
    for s in specs.synth_ca13(points_per_file=n, files=files):
        t = torch.empty(n * 12, dtype=torch.uint8, device=dev)
        ctx.synth_fill(s, 0, n, t.data_ptr(), None, stream)
        keep.append(t)
        cols.append(binding.make_columns(xyz=t.data_ptr(), n=n, scale=list(s.scale), offset=list(s.offset)))
        lmin, lmax = pkg.box_to_local(bmin, bmax, list(s.scale), list(s.offset))
        preds.append(pkg.Predicate.bounds(lmin, lmax))
    torch.cuda.synchronize()
    counter = torch.zeros(2, dtype=torch.int64, device=dev)
    cc = ctx.count_collector(device_counter=counter.data_ptr())
    ctx.set_option("k1_variant", 12)
    grids = [512, 576, 640, 672, 704, 736, 768, 800, 832, 864, 896, 960, 1024, 1280, 1536]
    times = {g: [] for g in grids}
    k = 0
    for r in range(14):
        for g in grids:
            ctx.set_option("k1_grid", g)
            f = k % files; k += 1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ctx.scan_dev(cols[f], preds[f], cc, stream); e1.record(); e1.synchronize()
            if r >= 2: times[g].append(e0.elapsed_time(e1))
    for g, t in times.items():
        t.sort(); med = t[len(t) // 2]
        print(f"grid {g:5d} ({g / 256:.2f} waves/CU): {n * 12 / med / 1e6:8.1f} GB/s  median {med:.4f} ms", flush=True)
    ctx.set_option("k1_grid", 0)
    cc.free()
