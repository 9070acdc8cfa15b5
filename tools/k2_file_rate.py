"""Developer tool: per-file class count (K2) — the one-wave pipelined kernel vs the 256-thread kernel, on 6
resident classification blocks of 163 M bytes visited round-robin (978 MB, beyond the Infinity Cache)."""
import os
os.environ.setdefault("PCQ_LAB", "1")  # the kernel shapes / microbenchmarks swept here live in libpcq_lab.so (make -C csrc lab)
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
binding = importlib.import_module("adhoc-queries-pointclouds_amd.binding")
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
n, files = 163_000_000, 6
dev = torch.device("cuda:0")
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); stream = ts.cuda_stream
with pkg.Context(0) as ctx:
    cols, keep = [], []
    for s in specs.synth_ca13(points_per_file=n, files=files):
        c = torch.empty(n + 64, dtype=torch.uint8, device=dev)
        ctx.synth_fill(s, 0, n, None, c.data_ptr() + 3, stream)  # odd alignment: the head/tail path runs too
        keep.append(c)
        cols.append(binding.make_columns(cls=c.data_ptr() + 3, n=n, scale=list(s.scale), offset=list(s.offset)))
    torch.cuda.synchronize()
    pred = pkg.Predicate.classification(6)
    counter = torch.zeros(2, dtype=torch.int64, device=dev)
    configs = [(0, b) for b in (2, 4)] + [(1, w) for w in (2, 3, 4, 5, 6, 8)]
    times = {c: [] for c in configs}
    ref = None
    for r in range(10):
        for pipe, w in configs:
            ctx.set_option("class_batch_pipe", pipe)
            ctx.set_option("class_batch_waves_per_cu" if pipe else "blocks_per_cu", w)
            counter.zero_()
            cc = ctx.count_collector(counter.data_ptr())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for c in cols:
                ctx.scan_dev(c, pred, cc, stream)
            e1.record(); e1.synchronize()
            got = int(counter[0].item())
            cc.free()
            ref = got if ref is None else ref
            assert got == ref, (pipe, w, got, ref)
            if r >= 2:
                times[(pipe, w)].append(e0.elapsed_time(e1))
    for (pipe, w), t in times.items():
        t.sort()
        med = t[len(t) // 2]
        print(f"{n * files / med / 1e6:8.1f} GB/s  median {med / files:.4f} ms/file  {'one-wave pipelined, %d waves/CU' % w if pipe else '256-thread kernel, blocks_per_cu %d' % w}", flush=True)
    print("count", ref)
