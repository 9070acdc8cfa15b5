"""Manual: lays out synthetic datasets the way the paper's driver expects them,
<root>/<dataset>/<extension>/ (run_query_experiments.rs:262-267), for adhoc-queries-pointclouds_amd/host/run_query_experiments.

Uses the oracle's generator and the test-side LAZER writer (hence under tests/).  Sizes are the §8(d)
recipes scaled down: --navvis-points (1 file), --doc-points / --ca13-points per file (8 / 16 files).

    python tests/manual/make_experiment_datasets.py ROOT [--navvis-points N] [--doc-points N] [--ca13-points N]
"""
import argparse
import importlib
import os
import shutil
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _oracle

specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--navvis-points", type=int, default=56_200_000)
    ap.add_argument("--doc-points", type=int, default=106_750_000 // 16)
    ap.add_argument("--ca13-points", type=int, default=163_000_000 // 32)
    ap.add_argument("--lazer-block", type=int, default=1_000_000)
    ap.add_argument("--formats", default="las,last,lazer")
    args = ap.parse_args()
    o = _oracle.Oracle()
    sets = {"navvis3": specs.synth_navvis(points_per_file=args.navvis_points),
            "doc": specs.synth_doc(points_per_file=args.doc_points),
            "ca13": specs.synth_ca13(points_per_file=args.ca13_points)}
    formats = args.formats.split(",")
    need = sum(int(s.n) * 34 for ss in sets.values() for s in ss) * len(formats)
    free = shutil.disk_usage(os.path.dirname(os.path.abspath(args.root)) or ".").free
    if free < need * 1.2:
        sys.exit(f"not enough disk space: need ~{need / 1e9:.1f} GB, free {free / 1e9:.1f} GB")
    t0 = time.time()
    total = 0
    for name, ss in sets.items():
        for ext in formats:
            os.makedirs(os.path.join(args.root, name, ext), exist_ok=True)
        for i, s in enumerate(ss):
            if "last" in formats or "lazer" in formats:
                image = o.synth_image(s, transposed=True, threads=8)
                if "last" in formats:
                    image.tofile(os.path.join(args.root, name, "last", f"{name}_{i:02d}.last"))
                    total += image.size
                if "lazer" in formats:
                    z = o.lazer_from_last(image, args.lazer_block, 4, 4)
                    z.tofile(os.path.join(args.root, name, "lazer", f"{name}_{i:02d}.lazer"))
                    total += z.size
                del image
            if "las" in formats:
                p = os.path.join(args.root, name, "las", f"{name}_{i:02d}.las")
                o.synth_write(s, p, transposed=False, threads=8)
                total += os.path.getsize(p)
        print(f"{name}: {len(ss)} files x {int(ss[0].n)} points, formats {formats}", flush=True)
    print(f"wrote {total / 1e9:.2f} GB in {time.time() - t0:.1f} s", flush=True)


if __name__ == "__main__":
    main()
