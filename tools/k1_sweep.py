"""Kernel-variant sweep for the count kernels on a real MI355X (developer tool, not the bench).

Times each K1 (bounds count) variant and K2 (class count) on a device-resident synthetic LAST file
with HIP events (torch.cuda.Event on the stream the kernels are launched on), interleaved rounds in
one process, and prints achieved algorithmic GB/s (12 B/point, 1 B/point).
"""
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
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--blocks", type=str, default="4,8,16")
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
        spec = specs.synth_ca13(points_per_file=n, files=1)[0]
        xyz = torch.empty(n * 12, dtype=torch.uint8, device=dev)
        cls = torch.empty(n, dtype=torch.uint8, device=dev)
        ctx.synth_fill(spec, 0, n, xyz.data_ptr(), cls.data_ptr(), stream)
        torch.cuda.synchronize()
        counter = torch.zeros(2, dtype=torch.int64, device=dev)
        cc = ctx.count_collector(device_counter=counter.data_ptr())
        bmin, bmax = specs.box("ca13_XL")
        lmin, lmax = pkg.box_to_local(bmin, bmax, list(spec.scale), list(spec.offset))
        cols = binding.make_columns(xyz=xyz.data_ptr(), cls=cls.data_ptr(), n=n, scale=list(spec.scale), offset=list(spec.offset))
        pb, pc = pkg.Predicate.bounds(lmin, lmax), pkg.Predicate.classification(6)
        results = {}
        for bpc in [int(b) for b in args.blocks.split(",")]:
            ctx.set_option("blocks_per_cu", bpc)
            for variant in (0, 1, 2, 3, "class"):
                if variant != "class":
                    ctx.set_option("k1_variant", variant)
                times = []
                for r in range(args.rounds + 2):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ctx.scan_dev(cols, pc if variant == "class" else pb, cc, stream)
                    e1.record()
                    e1.synchronize()
                    if r >= 2:
                        times.append(e0.elapsed_time(e1))
                times.sort()
                med = times[len(times) // 2]
                bpp = 1 if variant == "class" else 12
                results[(bpc, variant)] = med
                print(f"blocks/cu={bpc:2d} variant={variant!s:6} median {med:8.4f} ms  min {times[0]:8.4f} ms  "
                      f"{n * bpp / med / 1e6:9.1f} GB/s (min: {n * bpp / times[0] / 1e6:9.1f})", flush=True)
        ctx.set_option("k1_variant", 0)
        print("count check:", int(counter[0].item()))
        cc.free()


if __name__ == "__main__":
    main()
