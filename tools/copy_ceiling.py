"""What a plain device-to-device copy reaches on this GPU (read + write streams together): the ceiling the partition passes
of the grid collector are measured against.  Uses torch only as the plumbing for a copy kernel.  usage: copy_ceiling.py"""
import torch
for gb in (1, 4):
    n = gb * (1 << 30) // 4
    a = torch.empty(n, dtype=torch.int32, device="cuda").random_()
    b = torch.empty_like(a)
    for fn, name in ((lambda: b.copy_(a), "copy_"), (lambda: torch.add(a, 1, out=b), "add 1")):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{gb} GiB {name}: {ms:.3f} ms, read+write {2 * gb * 1.073741824 / ms:.2f} TB/s")
    del a, b
