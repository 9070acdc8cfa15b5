"""Turns gpurun_out/r03 (tools/r03_measure_a.sh, r03_measure_b.sh) into the round-3 evidence files under profiles/, and prints
the K1 figures of the committed rocprofv3 summary (DESIGN.md / profiles/README.md quote THAT line — round 2 quoted figures of a
summary that had since been regenerated).
usage: python tools/r03_make_profiles.py [git_head]"""
import csv, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(ROOT, "gpurun_out", "r03"), os.path.join(ROOT, "profiles")
head = sys.argv[1] if len(sys.argv) > 1 else subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
ALG_GB = 12 * 163_000_000 / 1e9


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def kstats(path, lim_us=5.0):
    out = []
    for r in csv.DictReader(open(path)):
        n = short(r["Name"])
        if any(x in n for x in ("at::", "elementwise", "rocprim", "synth", "index_", "vectorized", "copyBuffer", "fillBuffer")):
            continue
        if float(r["AverageNs"]) / 1e3 >= lim_us:
            out.append(f'  {n[:72]:72s} calls {r["Calls"]:>3s} avg {float(r["AverageNs"]) / 1e3:10.1f} us  min {float(r["MinNs"]) / 1e3:10.1f} us')
    return out


CASES = (("random", 100), ("random", 10), ("coherent", 100), ("coherent", 10))
ORDER = {"random": "generator order (uniform random inside the tile's box)",
         "coherent": "scan-strip order (COHERENT=10: strips 10 m wide in y and z, points sorted along x — the order of flight-line tiles)"}

# grid kernel stats, both point orders
if os.path.exists(f"{O}/final_random_100.log"):
    L = [f"# round 3 (final state, git {head}): grid collector per kernel, one synthetic ca13 file of 163 M points resident in HBM, query ca13_XL,",
         "# 4 repeats each (tools/grid_probe.py under rocprofv3 --kernel-trace --stats; tools/r03_measure_grid.sh).  scan = pass 0 (k_p0_part: ONE",
         "# reading of the points), asynchronous; count = the fold that the first accessor triggers (k_dir_transpose, k_bin_prefix, [k_bin_compact,]",
         "# [k_probe_distinct,] [k_level2,] k_fold / k_fold_dense, directory kernels, four synchronisations).  agg 0 = pass 0 folds a tile's duplicate",
         "# cells while that sheds a quarter of its matches; tuples = what the fold found pending.  The first repeat of a process pays the pool's",
         "# device allocations.", ""]
    for order, cell in CASES:
        L.append(f"== ca13_XL --density {cell}, {ORDER[order]} ==")
        L += ["  " + l.strip() for l in open(f"{O}/final_{order}_{cell}.log") if "cells" in l]
        L += kstats(f"{O}/prof_final_{order}_{cell}/g_kernel_stats.csv")
        L.append("")
    open(f"{P}/r03_grid_kernel_stats.txt", "w").write("\n".join(L))


def counters(order, cell):
    path = f"{O}/pmc_final_{order}_{cell}.txt"
    if not os.path.exists(path):
        return {}
    blocks, cur = {}, None
    for line in open(path).read().split("\n"):
        m = re.match(r"(.*) \((\d+) dispatches, per dispatch\)", line)
        if m:
            cur = blocks.setdefault(m.group(1).replace("void ", ""), {})
        elif cur is not None and "=" in line and not line.startswith("=="):
            for kv in line.split():
                k, v = kv.split("=")
                cur[k] = float(v)
    return blocks


if counters("random", 100):
    L = [f"# round 3 (final state, git {head}): HBM-side traffic of the grid collector's kernels, rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in SEPARATE",
         "# runs without tracing domains (tools/r03_grid_counters.sh), tools/grid_probe.py ca13_XL <cell> 163000000 2; per kernel and dispatch.",
         "# Units: counter value = KiB.  FETCH_SIZE counts a 128-byte request of a wide coalesced read as 64 bytes on gfx950 (MI355X_MICROARCH.md,",
         "# HBM section): the x2 column applies that correction — pass 0's reading of the positions and class bytes (2.12 GB) comes out at 2.06 GB with",
         "# it.  Algorithmic bytes of the query: 12 B x 163 M points = 1.956 GB.", ""]
    for order, cell in CASES:
        b = counters(order, cell)
        if not b:
            continue
        L.append(f"== ca13_XL --density {cell}, {order} order: GB per scan + fold ==")
        L.append(f"  {'kernel':50s} {'FETCH raw':>10s} {'FETCH x2':>10s} {'WRITE':>10s}")
        tf = tw = 0.0
        for k, c in b.items():
            f, w = c.get("FETCH_SIZE", 0) * 1024 / 1e9, c.get("WRITE_SIZE", 0) * 1024 / 1e9
            if f + w < 0.01 or any(x in k for x in ("synth", "at::", "rocprim", "elementwise")):
                continue
            L.append(f"  {k[:50]:50s} {f:10.3f} {2 * f:10.3f} {w:10.3f}")
            tf, tw = tf + f, tw + w
        L.append(f"  {'TOTAL':50s} {tf:10.3f} {2 * tf:10.3f} {tw:10.3f}")
        L.append(f"  traffic / algorithmic (1.956 GB): {(tf + tw) / ALG_GB:.1f} x (FETCH raw + WRITE) ... {(2 * tf + tw) / ALG_GB:.1f} x (FETCH x2 + WRITE)")
        L.append("")
    open(f"{P}/r03_grid_pmc.txt", "w").write("\n".join(L))
    L = [f"# round 3 (final state, git {head}): SQ counters of the grid collector's kernels (two --pmc passes of eight counters, no tracing domains;",
         "# tools/r03_grid_counters.sh), per dispatch.  2.55 M wave-instructions cover the file's 163 M points / tuples once:",
         "# SQ_INSTS_VALU / 2.55e6 = vector instructions per 64 points (tuples).  SQ_WAVE_CYCLES and the waits are sampled on a quarter of the waves.", ""]
    keep = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS",
            "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CYCLES")
    for order, cell in CASES:
        b = counters(order, cell)
        if not b:
            continue
        L.append(f"== ca13_XL --density {cell}, {order} order ==")
        for k, c in b.items():
            if c.get("SQ_INSTS_VALU", 0) < 1e7 or any(x in k for x in ("synth", "at::", "rocprim", "elementwise")):
                continue
            L.append("  " + k[:70])
            L.append("      " + "  ".join(f"{n}={c[n]:.3g}" for n in keep if n in c) + f"   VALU per 64 tuples = {c['SQ_INSTS_VALU'] / 2.55e6:.0f}")
        L.append("")
    open(f"{P}/r03_grid_sq_counters.txt", "w").write("\n".join(L))

# emit
if os.path.exists(f"{O}/emit_probe_ca13_XL.log"):
    L = [f"# round 3 (git {head}): buffer collector (stable emit of 31-byte records) on one 163 M-point ca13 file, tools/emit_probe.py under",
         "# rocprofv3 --kernel-trace --stats (tools/r03_emit_tiles.sh).  k_emit_points<KIND, RGB>: the colourless form no longer issues the three",
         "# masked colour loads per point, and a tile without a match returns before its first load (r03_emit_tiles.log).", ""]
    for q in ("ca13_XL", "ca13_S"):
        L.append(f"== {q} ==")
        L += ["  " + l.strip() for l in open(f"{O}/emit_probe_{q}.log") if "matches" in l]
        L += kstats(f"{O}/prof_emit_{q}/e_kernel_stats.csv")
        L.append("")
    open(f"{P}/r03_emit_probe.log", "w").write("\n".join(L))

for src, dst in (("bench_n1.json", "r03_bench_n1.json"), ("bench_torchrun_n1.json", "r03_bench_torchrun_n1.json"), ("bench_profiled.json", "r03_bench_n1_profiled_run.json"),
                 ("prof_bench/b_kernel_stats.csv", "r03_bench_n1_kernel_stats.csv"), ("prof_bench/b_kernel_trace.csv", "r03_bench_n1_kernel_trace.csv"),
                 ("collector_timings.log", "r03_collector_timings.log"), ("cli_e2e.log", "r03_cli_e2e.log")):
    if os.path.exists(f"{O}/{src}"):
        shutil.copy(f"{O}/{src}", f"{P}/{dst}")
if os.path.exists(f"{O}/pmc_latest.json"):
    pmc = json.load(open(f"{O}/pmc_latest.json"))
    pmc["git_head"] = head
    for dst in ("pmc_latest.json", "r03_pmc_traffic.json"):
        json.dump(pmc, open(f"{P}/{dst}", "w"), indent=1)
if os.path.exists(f"{O}/experiments/product.txt"):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "manual", "format_experiments.py"), f"{O}/experiments"], capture_output=True, text=True)
    if r.returncode == 0:
        open(f"{P}/r03_query_experiments.txt", "w").write(r.stdout)

if os.path.exists(f"{O}/emit_tiles.log"):
    L = [f"# round 3 (git {head}): buffer collector on one 163 M-point ca13 file, boxes that keep 100 % / 10 % / 1 % of the file's x range (FRAC), the file in",
         "# generator order (SORTED=0: every 2048-point tile holds a few matches) and sorted along x (SORTED=1: the matches are one run of the file, the",
         "# emit skips every other tile); tools/emit_probe.py through tools/r03_emit_tiles.sh.  The kernel lines at the end: SORTED=1 FRAC=0.1 under",
         "# rocprofv3 --kernel-trace --stats.  GB/s = algorithmic bytes (13 B read per point + 31 B written per match) / wall time.", ""]
    L += [l.rstrip() for l in open(f"{O}/emit_tiles.log") if "amdgpu.ids" not in l]
    open(f"{P}/r03_emit_tiles.log", "w").write("\n".join(L) + "\n")

# the K1 line, from the committed summary
ks = f"{P}/r03_bench_n1_kernel_stats.csv"
if os.path.exists(ks):
    for r in csv.DictReader(open(ks)):
        if "k_bounds_count_batch_pipe" in r["Name"]:
            avg, mn, calls = float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, int(r["Calls"])
            gb = 12 * 2_608_000_000 / 1e9
            print(f"K1 under rocprofv3 ({os.path.basename(ks)}): {calls} launches, average {avg:.3f} ms, min {mn:.3f} ms -> "
                  f"{gb / (avg * 1e-3):.0f} GB/s = {gb / (avg * 1e-3) / 8000:.3f} of 8 TB/s (min: {gb / (mn * 1e-3) / 8000:.3f})")
print("profiles written for", head)
