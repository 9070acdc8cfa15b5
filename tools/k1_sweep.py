"""Kernel-variant sweep for the count kernels on a real MI355X (developer tool, not the bench).

Times each K1 (bounds count) variant, the batched K1 and K2 (class count) with HIP events
(torch.cuda.Event on the stream the kernels are launched on), interleaved rounds in one process, and
prints achieved algorithmic GB/s (12 B/point, 1 B/point).

The working set rotates over `--files` device-resident synthetic files (default 8 x 163 M points =
15.6 GB) so that no launch re-reads data the 256 MiB Infinity Cache could still hold: a sweep that
re-reads ONE 2 GB file reports ~10 % more than the HBM stream really delivers.
"""
import os
os.environ.setdefault("PCQ_LAB", "1")  # the kernel shapes / microbenchmarks swept here live in libpcq_lab.so (make -C csrc lab)
import argparse
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
binding = importlib.import_module("adhoc-queries-pointclouds_amd.binding")
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=163_000_000)
    ap.add_argument("--files", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=16)
    ap.add_argument("--blocks", type=str, default="2,3,4,6,8")
    ap.add_argument("--variants", type=str, default="0,3,5,1")
    args = ap.parse_args()
    n = args.points
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    tstream = torch.cuda.Stream()  # a real (non-null) stream: handle 0 would mean "the context's stream"
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    assert stream != 0
    with pkg.Context(0) as ctx:
        print(json.dumps(ctx.device_info()))
        ss = specs.synth_ca13(points_per_file=n, files=args.files)
        xyz, cls, cols = [], [], []
        bmin, bmax = specs.box("ca13_XL")
        preds = []
        for s in ss:
            x = torch.empty(n * 12, dtype=torch.uint8, device=dev)
            c = torch.empty(n, dtype=torch.uint8, device=dev)
            ctx.synth_fill(s, 0, n, x.data_ptr(), c.data_ptr(), stream)
            xyz.append(x)
            cls.append(c)
            cols.append(binding.make_columns(xyz=x.data_ptr(), cls=c.data_ptr(), n=n, scale=list(s.scale), offset=list(s.offset)))
            lmin, lmax = pkg.box_to_local(bmin, bmax, list(s.scale), list(s.offset))
            preds.append(pkg.Predicate.bounds(lmin, lmax))
        torch.cuda.synchronize()
        counter = torch.zeros(2, dtype=torch.int64, device=dev)
        cc = ctx.count_collector(device_counter=counter.data_ptr())
        pc = pkg.Predicate.classification(6)
        cpreds = [pc] * args.files
        variants = [int(v) for v in args.variants.split(",")]
        configs = [(bpc, v) for bpc in [int(b) for b in args.blocks.split(",")] for v in variants + ["batch", "batchw1", "batchw1x3", "batchpipe", "class", "cbatch"]]
        times = {c: [] for c in configs}
        k = 0
        for r in range(args.rounds + 2):  # interleaved rounds in one process (cdna guide rule 24)
            for bpc, variant in configs:
                ctx.set_option("blocks_per_cu", min(bpc, 16))
                ctx.set_option("k1_waves_per_cu", bpc)  # variants 8..11: single-wave workgroups per CU
                ctx.set_option("k1_variant", variant if isinstance(variant, int) else 0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ctx.set_option("batch_variant", {"batchw1": 1, "batchw1x3": 2, "batchpipe": 3}.get(variant, 0))
                if variant in ("batchw1", "batchw1x3", "batchpipe"):  # one wave per workgroup, 2 / 3 tiles per step; blocks/cu = waves per CU here
                    ctx.set_option("batch_waves_per_cu", bpc)
                if variant in ("batch", "batchw1", "batchw1x3", "batchpipe"):
                    e0.record()
                    ctx.scan_dev_count_batch(cols, preds, counter.data_ptr(), stream)
                    e1.record()
                elif variant == "cbatch":
                    e0.record()
                    ctx.scan_dev_count_batch(cols, cpreds, counter.data_ptr(), stream)
                    e1.record()
                else:
                    f = k % args.files
                    k += 1
                    e0.record()
                    ctx.scan_dev(cols[f], pc if variant == "class" else preds[f], cc, stream)
                    e1.record()
                e1.synchronize()
                if r >= 2:
                    times[(bpc, variant)].append(e0.elapsed_time(e1))
        for (bpc, variant), t in times.items():
            t.sort()
            med = t[len(t) // 2]
            nbytes = n * (1 if variant in ("class", "cbatch") else 12) * (args.files if variant in ("batch", "batchw1", "batchw1x3", "batchpipe", "cbatch") else 1)
            print(f"blocks/cu={bpc:2d} variant={variant!s:9} median {med:8.4f} ms  min {t[0]:8.4f} ms  "
                  f"{nbytes / med / 1e6:9.1f} GB/s (min-time: {nbytes / t[0] / 1e6:9.1f})", flush=True)
        print("count check:", int(counter[0].item()))
        cc.free()


if __name__ == "__main__":
    main()
