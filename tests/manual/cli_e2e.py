"""End-to-end wall clock of the `query` CLI (files on disk, page cache warm) next to the oracle CLI
(the C restatement of the reference's `query --optimized --parallel`; the oracle CLI is single-threaded,
the multi-threaded oracle number is bench.py's cpu_baseline).  Manual check kept under tests/ because it
uses the oracle; the numbers go to DESIGN.md, not to the bench line.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle  # noqa: E402  (file generator)

specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
QUERY = os.path.join(ROOT, "adhoc-queries-pointclouds_amd", "host", "query")
ORACLE = os.path.join(ROOT, "oracle", "query_oracle")


ALL_TIMES = {}
SETTLE = 1.0  # seconds in front of every run: a GPU process started right behind another one waits 0.1-0.2 s in hsa_init for the
              # driver's teardown of its predecessor (profiles/r03_hip_startup_env.log); the reference's protocol has `sync; purge` here


def run(exe, args, repeat=3, tag=None):
    best, out, times = None, None, []
    for _ in range(repeat):
        time.sleep(SETTLE)
        t0 = time.perf_counter()
        r = subprocess.run([exe] + args, capture_output=True, text=True, env=dict(os.environ, PCQ_EXIT=os.environ.get("PCQ_EXIT", "fast")))
        dt = time.perf_counter() - t0
        if r.returncode != 0:
            raise SystemExit(f"{exe} failed: {r.stderr}")
        times.append(round(dt, 4))
        best = dt if best is None or dt < best else best
        out = [l for l in r.stdout.splitlines() if not l.startswith("Searched")]
    if tag:
        ALL_TIMES[tag] = times
    return best, out


def phases(exe, args):
    """One more run with PCQ_TIMING=1: the per-phase lines (plans, context ready, per file, merge) of the CLI's stderr."""
    time.sleep(SETTLE)
    r = subprocess.run([exe] + args, capture_output=True, text=True, env=dict(os.environ, PCQ_TIMING="1", PCQ_EXIT=os.environ.get("PCQ_EXIT", "fast")))
    lines = [l for l in r.stderr.splitlines() if l.startswith("[pcq]") or l.startswith("pcq:")]
    files = [float(l.split(" searched in ")[1].split(" ms")[0]) for l in lines if " searched in " in l]
    keep = [l for l in lines if " searched in " not in l and "pool block" not in l]
    if files:
        keep.append("[pcq] %d files searched, per file min %.1f / median %.1f / max %.1f ms, sum %.1f ms" % (
            len(files), min(files), sorted(files)[len(files) // 2], max(files), sum(files)))
    return keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=16)
    ap.add_argument("--points", type=int, default=20_000_000)
    ap.add_argument("--threads-per-gpu", type=int, default=1)
    ap.add_argument("--settle", type=float, default=1.0, help="seconds in front of every run (0: back to back)")
    args = ap.parse_args()
    global SETTLE
    SETTLE = args.settle
    o = _oracle.Oracle()
    d = tempfile.mkdtemp(prefix="pcq_e2e_", dir="/tmp")
    ss = specs.synth_ca13(points_per_file=args.points, files=args.files)
    for i, s in enumerate(ss):
        o.synth_write(s, os.path.join(d, f"tile{i:02d}.last"), threads=32)
    total_pts = args.files * args.points
    res = {"files": args.files, "points": total_pts, "bytes": sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))}
    xl = "643431.76;3883547.565;-46194.145;736910.93;3977026.735;47285.025"
    for name, q in (("bounds_XL", ["--bounds", xl]), ("bounds_S", ["--bounds", "665000;3910000;0;705000;3950000;480"]),
                    ("class_6", ["--class", "6"]), ("bounds_XL_density_100", ["--bounds", xl, "--density", "100"]),
                    ("bounds_XL_density_10", ["--bounds", xl, "--density", "10"])):
        base = ["-i", d, "--optimized", "--parallel"] + q
        t_gpu, out_gpu = run(QUERY, base + ["--threads-per-gpu", str(args.threads_per_gpu)], repeat=5, tag=name)
        for l in phases(QUERY, base + ["--threads-per-gpu", str(args.threads_per_gpu)]):
            print("   ", name, l, flush=True)
        t_gpu2, _ = run(QUERY, base + ["--threads-per-gpu", "2"])
        t_gpu8, _ = run(QUERY, base + ["--threads-per-gpu", "4"])
        t_cpu, out_cpu = run(ORACLE, base, repeat=1)
        assert sorted(out_gpu) == sorted(out_cpu), (out_gpu, out_cpu)
        res[name] = {"gpu_cli_s": t_gpu, "gpu_cli_s_2thr": t_gpu2, "gpu_cli_s_4thr": t_gpu8, "oracle_cli_s": t_cpu, "gpu_Mpts_per_s": total_pts / t_gpu / 1e6,
                     "oracle_cli_Mpts_per_s": total_pts / t_cpu / 1e6, "stdout": out_gpu[-1] if out_gpu else ""}
        res[name]["gpu_cli_all_runs_s"] = ALL_TIMES.get(name)
        print(name, json.dumps(res[name]), flush=True)
    for f in os.listdir(d):
        os.remove(os.path.join(d, f))
    os.rmdir(d)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
