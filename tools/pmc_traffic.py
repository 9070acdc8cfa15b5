"""Collects HBM traffic of the dominant bench kernel from PMC counters (run on the GPU box).

Per /opt/skills/guides/MI355X_MICROARCH.md §HBM and cdna_hip_programming.md §7:
  * counters in their own runs, FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC has 4 slots;
    FETCH_SIZE costs 3, WRITE_SIZE 2), no tracing domains combined with --pmc;
  * unit: KiB -> bytes = value * 1024;
  * gfx950 correction: FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced
    streaming read -> doubled.  WRITE_SIZE is exact for 16 B/lane streaming stores.
Writes a JSON summary (per launch of the kernel, averaged over the profiled launches).
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_pass(counter, outdir, bench_args):
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", outdir, "--", "python3",
           os.path.join(ROOT, "bench.py")] + bench_args
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-2000:] + r.stderr[-2000:])
        raise SystemExit(f"rocprofv3 --pmc {counter} failed")
    rows = []
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows, r.stdout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pmc.json"))
    ap.add_argument("--kernel", default="k_bounds_count_batch_pipe<2>")
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    bench_args = ["--steps", str(args.steps), "--warmup", "1", "--no-cpu-baseline"]
    summary = {"kernel": args.kernel, "bench_args": bench_args, "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads), KiB -> bytes"}
    bench_line = None
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        rows, out = run_pass(counter, os.path.join(ROOT, "gpurun_out", "pmc_" + counter.lower()), bench_args)
        for line in out.splitlines():
            if line.startswith("{") and '"metric"' in line:
                bench_line = json.loads(line)
        hits = [r for r in rows if args.kernel in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter]
        vals = [float(r["Counter_Value"]) for r in hits]
        names = sorted({r["Kernel_Name"] for r in hits})
        if len(names) != 1:
            raise SystemExit(f"--kernel {args.kernel!r} matches {len(names)} kernels in the {counter} pass: {names}")
        summary["kernel_name"] = names[0]  # exactly as the counter CSV has it
        summary[counter + "_raw_kib_per_launch"] = sum(vals) / len(vals) if vals else None
        summary[counter + "_launches"] = len(vals)
    f, w = summary.get("FETCH_SIZE_raw_kib_per_launch"), summary.get("WRITE_SIZE_raw_kib_per_launch")
    if f is not None and w is not None:
        summary["hbm_read_bytes_per_launch"] = 2.0 * f * 1024.0
        summary["hbm_write_bytes_per_launch"] = w * 1024.0
        summary["hbm_bytes_per_launch"] = summary["hbm_read_bytes_per_launch"] + summary["hbm_write_bytes_per_launch"]
    summary["git_head"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or "unknown"
    if bench_line:
        rl = bench_line["roofline"]
        summary["kernel_source_id"] = rl.get("kernel_source_id")
        if rl.get("kernel") not in summary.get("kernel_name", ""):
            raise SystemExit(f"bench.py timed {rl.get('kernel')!r}, the counters are of {summary.get('kernel_name')!r}")
        summary["algorithmic_bytes_per_launch"] = rl["algorithmic_bytes_per_launch"]
        summary["points_per_launch"] = int(round(rl["algorithmic_bytes_per_launch"] / 12))
        if summary.get("hbm_bytes_per_launch"):
            summary["traffic_over_algorithmic"] = summary["hbm_bytes_per_launch"] / rl["algorithmic_bytes_per_launch"]
    json.dump(summary, open(args.out, "w"), indent=1)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
