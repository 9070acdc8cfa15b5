"""Developer tool: batched class count (K2) — the 256-thread kernel vs one-wave workgroups with 4-12 KiB per
step, by waves per CU; 16 classification blocks of 163 M bytes resident (2.6 GB, beyond the Infinity Cache)."""
import os
os.environ.setdefault("PCQ_LAB", "1")  # the kernel shapes / microbenchmarks swept here live in libpcq_lab.so (make -C csrc lab)
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
binding = importlib.import_module("adhoc-queries-pointclouds_amd.binding")
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
n, files = 163_000_000, 16
dev = torch.device("cuda:0")
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); stream = ts.cuda_stream
with pkg.Context(0) as ctx:
    cols = []
    keep = []
    for s in specs.synth_ca13(points_per_file=n, files=files):
        c = torch.empty(n, dtype=torch.uint8, device=dev)
        ctx.synth_fill(s, 0, n, None, c.data_ptr(), stream)
        keep.append(c)
        cols.append(binding.make_columns(cls=c.data_ptr(), n=n, scale=list(s.scale), offset=list(s.offset)))
    torch.cuda.synchronize()
    preds = [pkg.Predicate.classification(6)] * files
    counter = torch.zeros(2, dtype=torch.int64, device=dev)
    configs = [(0, b, 0) for b in (2, 3, 4, 6)] + [(l, w, 0) for l in (4, 6, 8, 12) for w in (4, 5, 6, 8, 10, 12, 16)] + \
              [(l, w, 1) for l in (4, 6, 8, 12) for w in (2, 3, 4, 5, 6, 8)]
    times = {c: [] for c in configs}
    for r in range(12):
        for loads, w, pipe in configs:
            ctx.set_option("class_batch_loads", loads)
            ctx.set_option("class_batch_pipe", pipe)
            if loads:
                ctx.set_option("class_batch_waves_per_cu", w)
            else:
                ctx.set_option("blocks_per_cu", w)
            counter.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.scan_dev_count_batch(cols, preds, counter.data_ptr(), stream)
            e1.record(); e1.synchronize()
            if r >= 2:
                times[(loads, w, pipe)].append(e0.elapsed_time(e1))
        if r == 0:
            first = int(counter[0].item())
        assert int(counter[0].item()) == first
    for (loads, w, pipe), t in times.items():
        t.sort()
        med = t[len(t) // 2]
        what = f"one-wave workgroups{', pipelined' if pipe else ''}, {loads:2d} KiB per step, {w:2d} waves/CU ({loads * w:3d}{'+' if pipe else ''} KiB in flight/CU)" if loads else f"256-thread kernel, blocks_per_cu option {w}"
        print(f"{n * files / med / 1e6:8.1f} GB/s  median {med:.4f} ms  {what}", flush=True)
    print("count", first)
