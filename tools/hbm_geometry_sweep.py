"""Developer tool: read-only stream rate vs launch geometry (waves per CU, loads in flight per wave)."""
import os
os.environ.setdefault("PCQ_LAB", "1")  # the kernel shapes / microbenchmarks swept here live in libpcq_lab.so (make -C csrc lab)
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
n_bytes = 1_956_000_000
nbuf = 8
dev = torch.device("cuda:0")
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); stream = ts.cuda_stream
with pkg.Context(0) as ctx:
    bufs = [torch.randint(0, 255, (n_bytes,), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    torch.cuda.synchronize()
    configs = []
    for loads in (1, 2, 3, 4, 6, 8):
        for threads in (64, 128, 256, 512):
            for wpc in (4, 6, 8, 10, 12, 16, 24):  # waves per CU
                waves_per_block = threads // 64
                if (wpc * 256) % waves_per_block:
                    continue
                configs.append((loads, threads, wpc * 256 // waves_per_block, wpc))
    times = {c: [] for c in configs}
    k = 0
    for r in range(7):
        for cfg in configs:
            loads, threads, blocks, wpc = cfg
            b = bufs[k % nbuf]; k += 1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = ctx.lib.pcq_membench_read_tiles(ctx.handle, C.c_void_p(b.data_ptr()), n_bytes, loads, threads, blocks, C.c_void_p(stream))
            assert rc == 0
            e1.record(); e1.synchronize()
            if r >= 2: times[cfg].append(e0.elapsed_time(e1))
    rows = []
    for cfg, t in times.items():
        t.sort()
        rows.append((n_bytes / t[len(t) // 2] / 1e6, cfg))
    for gbs, (loads, threads, blocks, wpc) in sorted(rows, reverse=True)[:25]:
        print(f"{gbs:8.1f} GB/s  loads/tile={loads} threads={threads} blocks={blocks} waves/CU={wpc} KiB in flight/CU={loads * wpc}")
    print("...")
    for gbs, (loads, threads, blocks, wpc) in sorted(rows, reverse=True)[-5:]:
        print(f"{gbs:8.1f} GB/s  loads/tile={loads} threads={threads} blocks={blocks} waves/CU={wpc} KiB in flight/CU={loads * wpc}")
