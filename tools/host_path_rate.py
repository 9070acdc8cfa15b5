"""PCIe-inclusive rate of the host-block path (pcq_scan_host): positions in ordinary host memory
(as an mmapped, page-cache-warm file would be) -> pinned staging -> hipMemcpyAsync -> K1.
Reported in DESIGN.md next to (never instead of) the HBM-resident bench value.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
binding = importlib.import_module("adhoc-queries-pointclouds_amd.binding")
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=100_000_000)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    n = args.points
    spec = specs.synth_ca13(points_per_file=n, files=1)[0]
    with pkg.Context(0) as ctx:
        # generate on the device, copy to plain (pageable) host memory
        d = ctx.alloc(12 * n)
        ctx.synth_fill(spec, 0, n, d, None)
        host = np.empty(3 * n, dtype=np.int32)
        ctx.to_host(host, d)
        ctx.free(d)
        bmin, bmax = specs.box("ca13_XL")
        lmin, lmax = pkg.box_to_local(bmin, bmax, list(spec.scale), list(spec.offset))
        cols = binding.make_columns(xyz=host.ctypes.data, n=n, scale=list(spec.scale), offset=list(spec.offset))
        pred = pkg.Predicate.bounds(lmin, lmax)
        out = {}
        for threads, chunk in ((1, 2 << 20), (2, 2 << 20), (4, 1 << 20), (4, 2 << 20), (4, 8 << 20), (6, 2 << 20), (8, 2 << 20), (8, 8 << 20)):
            ctx.set_option("chunk_points", chunk)
            ctx.set_option("copy_threads", threads)
            times = []
            for _ in range(args.rounds + 1):
                cc = ctx.count_collector()
                t0 = time.perf_counter()
                ctx.scan_host(cols, pred, cc)
                cnt = cc.point_count()
                times.append(time.perf_counter() - t0)
                cc.free()
            times = sorted(times[1:])
            med = times[len(times) // 2]
            out[f"threads_{threads}_chunk_{chunk}"] = {"seconds": med, "mpoints_per_s": n / med / 1e6, "gb_per_s": 12 * n / med / 1e9, "count": cnt}
        print(json.dumps({"points": n, "path": "pageable host memory -> pinned staging (memcpy) -> hipMemcpyAsync -> K1", **out}))


if __name__ == "__main__":
    main()
