"""Turns gpurun_out/r02 (tools/r02_measure_a.sh, r02_measure_b.sh) into the round-2 evidence files under profiles/.
usage: python tools/r02_make_profiles.py [git_head]"""
import csv, glob, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(ROOT, "gpurun_out", "r02"), os.path.join(ROOT, "profiles")
head = sys.argv[1] if len(sys.argv) > 1 else subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()

def short(n):
    return n.replace("(anonymous namespace)::", "").split("(")[0]

def kstats(path, lim_us=1.0):
    out = []
    for r in csv.DictReader(open(path)):
        if float(r["AverageNs"]) / 1e3 >= lim_us:
            out.append(f'  {short(r["Name"])[:72]:72s} calls {r["Calls"]:>3s} avg {float(r["AverageNs"]) / 1e3:10.1f} us  min {float(r["MinNs"]) / 1e3:10.1f} us')
    return out

# grid kernel stats
L = ["# round 2 (final state, git %s): grid collector per kernel, one synthetic ca13 file of 163 M points resident in HBM, query ca13_XL, 4 repeats each" % head,
     "# (tools/grid_probe.py under rocprofv3 --kernel-trace --stats; tools/r02_measure_b.sh).  scan = pass 0 (k_p0_hist, two small scans,",
     "# k_p0_scatter), asynchronous; count = the fold that the first accessor triggers (probe, [k_level2,] k_fold / k_fold_dense, directory kernels).",
     "# The partition kernels move by +-15 % from GPU box to GPU box on the same code: see r02_grid_progress.txt.", ""]
for cell in (100, 10):
    L.append(f"== ca13_XL --density {cell} ==")
    L += ["  " + l.strip() for l in open(f"{O}/grid_probe_{cell}.log") if "cells" in l]
    L += kstats(f"{O}/prof_grid_{cell}/g_kernel_stats.csv")
    L.append("")
open(f"{P}/r02_grid_kernel_stats.txt", "w").write("\n".join(L))

# counters
def counters(cell):
    txt = open(f"{O}/pmc_final_{cell}.txt").read()
    blocks, cur = {}, None
    for line in txt.split("\n"):
        m = re.match(r"(.*) \((\d+) dispatches, per dispatch\)", line)
        if m:
            cur = blocks.setdefault(m.group(1).replace("void ", ""), {})
        elif cur is not None and "=" in line and not line.startswith("=="):
            for kv in line.split():
                k, v = kv.split("=")
                cur[k] = float(v)
    return blocks
L = ["# round 2 (final state, git %s): HBM-side traffic of the grid collector's kernels, rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in SEPARATE" % head,
     "# runs without tracing domains (tools/r02_grid_counters.sh), tools/grid_probe.py ca13_XL <cell> 163000000 2; per kernel and dispatch.",
     "# Units: counter value = KiB.  FETCH_SIZE counts a 128-byte request of a wide coalesced read as 64 bytes on gfx950 (MI355X_MICROARCH.md,",
     "# HBM section): the x2 column applies that correction; for the narrower reads of these kernels the truth lies between raw and x2.",
     "# WRITE_SIZE counts every partial-line write request at its request size: the scattered runs of the partition kernels read higher",
     "# than the bytes they store.  Algorithmic bytes of the query: 12 B x 163 M points = 1.956 GB.", ""]
for cell in (100, 10):
    b = counters(cell)
    L.append(f"== ca13_XL --density {cell}: GB per scan + fold ==")
    L.append(f"  {'kernel':50s} {'FETCH raw':>10s} {'FETCH x2':>10s} {'WRITE':>10s}")
    tf = tw = 0.0
    for k, c in b.items():
        f, w = c.get("FETCH_SIZE", 0) * 1024 / 1e9, c.get("WRITE_SIZE", 0) * 1024 / 1e9
        if f + w < 0.01 or "synth" in k:
            continue
        L.append(f"  {k[:50]:50s} {f:10.3f} {2 * f:10.3f} {w:10.3f}")
        tf, tw = tf + f, tw + w
    L.append(f"  {'TOTAL':50s} {tf:10.3f} {2 * tf:10.3f} {tw:10.3f}")
    L.append(f"  traffic / algorithmic (1.956 GB): {(tf + tw) / 1.956:.1f} x (FETCH raw + WRITE) ... {(2 * tf + tw) / 1.956:.1f} x (FETCH x2 + WRITE)")
    L.append("")
open(f"{P}/r02_grid_pmc.txt", "w").write("\n".join(L))
L = ["# round 2 (final state, git %s): SQ counters of the grid collector's kernels (two --pmc passes of eight counters, no tracing domains;" % head,
     "# tools/r02_grid_counters.sh), per dispatch.  2.55 M wave-instructions cover the file's 163 M points / tuples once:",
     "# SQ_INSTS_VALU / 2.55e6 = vector instructions per 64 tuples.", ""]
keep = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES")
for cell in (100, 10):
    L.append(f"== ca13_XL --density {cell} ==")
    for k, c in counters(cell).items():
        if c.get("SQ_INSTS_VALU", 0) < 1e7 or "synth" in k:
            continue
        L.append("  " + k[:70])
        L.append("      " + "  ".join(f"{n}={c[n]:.3g}" for n in keep if n in c) + f"   VALU per 64 tuples = {c['SQ_INSTS_VALU'] / 2.55e6:.0f}")
    L.append("")
open(f"{P}/r02_grid_sq_counters.txt", "w").write("\n".join(L))

# emit
L = ["# round 2 (final state, git %s): buffer collector (stable emit of 31-byte records) on one 163 M-point ca13 file, tools/emit_probe.py under" % head,
     "# rocprofv3 --kernel-trace --stats.  scan = k_tile_counts + three scan kernels + k_emit_points, asynchronous, timed to the synchronise.", ""]
for q in ("ca13_XL", "ca13_S"):
    L.append(f"== {q} ==")
    L += ["  " + l.strip() for l in open(f"{O}/emit_probe_{q}.log") if "matches" in l]
    L += kstats(f"{O}/prof_emit_{q}/e_kernel_stats.csv")
    L.append("")
open(f"{P}/r02_emit_probe.log", "w").write("\n".join(L))

for src, dst in (("bench_n1.json", "r02_bench_n1.json"), ("bench_torchrun_n1.json", "r02_bench_torchrun_n1.json"), ("bench_profiled.json", "r02_bench_n1_profiled_run.json"),
                 ("prof_bench/b_kernel_stats.csv", "r02_bench_n1_kernel_stats.csv"), ("prof_bench/b_kernel_trace.csv", "r02_bench_n1_kernel_trace.csv"),
                 ("collector_timings.log", "r02_collector_timings.log"), ("resident_rate.log", "r02_resident_rate.log"), ("copy_ceiling.log", "r02_copy_ceiling.log")):
    if os.path.exists(f"{O}/{src}"):
        shutil.copy(f"{O}/{src}", f"{P}/{dst}")
pmc = json.load(open(f"{O}/pmc_latest.json"))
pmc["git_head"] = head
for dst in ("pmc_latest.json", "r02_pmc_traffic.json"):
    json.dump(pmc, open(f"{P}/{dst}", "w"), indent=1)
print("profiles written for", head)
