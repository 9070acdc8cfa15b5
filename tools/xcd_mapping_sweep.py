"""Developer tool: does an XCD-aware workgroup -> tile mapping change the read-stream rate?
Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  K1 streams every byte once, so
there is no reuse for an L2 to capture; this measures whether locality per XCD matters anyway."""
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
NAMES = {0: "strided tiles (K1)", 1: "XCD-contiguous inside each window", 2: "one contiguous eighth per XCD"}
with pkg.Context(0) as ctx:
    bufs = [torch.randint(0, 255, (n_bytes,), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    torch.cuda.synchronize()
    configs = [(m, threads, bpc) for m in (0, 1, 2) for threads in (256, 512) for bpc in (1, 2, 3, 4)]
    times = {c: [] for c in configs}
    k = 0
    for r in range(9):
        for cfg in configs:
            m, threads, bpc = cfg
            b = bufs[k % nbuf]; k += 1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = ctx.lib.pcq_membench_read_xcd(ctx.handle, C.c_void_p(b.data_ptr()), n_bytes, m, threads, 256 * bpc, C.c_void_p(stream))
            assert rc == 0
            e1.record(); e1.synchronize()
            if r >= 2: times[cfg].append(e0.elapsed_time(e1))
    for (m, threads, bpc), t in times.items():
        t.sort()
        print(f"{n_bytes / t[len(t) // 2] / 1e6:8.1f} GB/s  mapping {m} ({NAMES[m]}), threads={threads}, blocks/CU={bpc}, KiB in flight/CU={3 * threads // 64 * bpc}")
