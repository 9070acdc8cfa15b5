"""Turns gpurun_out/r04 (tools/r04_measure_a.sh, tools/r04_final_grid.sh) into the round-4 evidence files under profiles/, and prints
the K1 figures of the committed rocprofv3 summary (DESIGN.md / profiles/README.md quote THAT line).
usage: python tools/r04_make_profiles.py [git_head]"""
import csv, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(ROOT, "gpurun_out", "r04"), os.path.join(ROOT, "profiles")
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
if os.path.exists(f"{O}/fin_random_100.log"):
    L = [f"# round 4 (final state, git {head}): grid collector per kernel, one synthetic ca13 file of 163 M points resident in HBM, query ca13_XL,",
         "# 4 repeats each (tools/grid_probe.py under rocprofv3 --kernel-trace --stats; tools/r04_measure_grid.sh).  scan = pass 0 (k_p0_part: ONE",
         "# reading of the points), asynchronous; count = the fold that the first accessor triggers (k_dir_transpose, k_bin_prefix, [k_probe_distinct,]",
         "# [k_level2,] k_fold_stream / k_fold_dense [/ k_fold], directory kernels, two synchronisations).  agg 0 = pass 0 folds a tile's duplicate",
         "# cells while that sheds a quarter of its matches; tuples = what the fold found pending.  The first repeat of a process pays the pool's",
         "# device allocations.", ""]
    for order, cell in CASES:
        L.append(f"== ca13_XL --density {cell}, {ORDER[order]} ==")
        L += ["  " + l.strip() for l in open(f"{O}/fin_{order}_{cell}.log") if "cells" in l]
        L += kstats(f"{O}/prof_fin_{order}_{cell}/g_kernel_stats.csv")
        L.append("")
    if os.path.exists(f"{O}/final_unprofiled.txt"):
        L += [l.rstrip() for l in open(f"{O}/final_unprofiled.txt")]
    open(f"{P}/r04_grid_kernel_stats.txt", "w").write("\n".join(L) + "\n")


def counters(order, cell):
    path = f"{O}/pmc_fin_{order}_{cell}.txt"
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
    L = [f"# round 4 (final state, git {head}): HBM-side traffic of the grid collector's kernels, rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in SEPARATE",
         "# runs without tracing domains (tools/r04_grid_counters.sh), tools/grid_probe.py ca13_XL <cell> 163000000 2; per kernel and dispatch.",
         "# Units: counter value = KiB.  FETCH_SIZE counts a 128-byte request of a wide coalesced read as 64 bytes on gfx950 (MI355X_MICROARCH.md,",
         "# HBM section): FETCH x 2 is the reading quoted everywhere (pass 0's reading of the positions and class bytes, 2.12 GB, comes out at",
         "# 2.0-2.1 GB with it).  Algorithmic bytes of the query: 12 B x 163 M points = 1.956 GB.", ""]
    for order, cell in CASES:
        b = counters(order, cell)
        if not b:
            continue
        L.append(f"== ca13_XL --density {cell}, {order} order: GB per scan + fold ==")
        L.append(f"  {'kernel':50s} {'FETCH x2':>10s} {'WRITE':>10s}")
        tf = tw = 0.0
        for k, c in b.items():
            f, w = c.get("FETCH_SIZE", 0) * 1024 / 1e9, c.get("WRITE_SIZE", 0) * 1024 / 1e9
            if f + w < 0.01 or any(x in k for x in ("synth", "at::", "rocprim", "elementwise")):
                continue
            L.append(f"  {k[:50]:50s} {2 * f:10.3f} {w:10.3f}")
            tf, tw = tf + f, tw + w
        L.append(f"  {'TOTAL':50s} {2 * tf:10.3f} {tw:10.3f}")
        L.append(f"  traffic / algorithmic (1.956 GB): {(2 * tf + tw) / ALG_GB:.1f} x")
        L.append("")
    open(f"{P}/r04_grid_pmc.txt", "w").write("\n".join(L))
    L = [f"# round 4 (final state, git {head}): SQ counters of the grid collector's kernels (two --pmc passes of eight counters, no tracing domains;",
         "# tools/r04_grid_counters.sh), per dispatch.  2.55 M wave-instructions cover the file's 163 M points / tuples once:",
         "# SQ_INSTS_VALU / 2.55e6 = vector instructions per 64 points (tuples).  A wave64 vector instruction occupies its 16-lane SIMD for four cycles:",
         "# issue = SQ_INSTS_VALU x 4 / (1024 SIMDs x kernel cycles) — e.g. 5.2e8 x 4 / 1024 / (1.11 ms x 2.4 GHz) = 0.76 for k_fold_stream.  SQ_WAVE_CYCLES and",
         "# the waits are in units of four cycles.", ""]
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
    open(f"{P}/r04_grid_sq_counters.txt", "w").write("\n".join(L))

# emit
if os.path.exists(f"{O}/emit_probe_ca13_XL.log"):
    L = [f"# round 4 (git {head}): buffer collector (stable emit of 31-byte records) on one 163 M-point ca13 file, tools/emit_probe.py under",
         "# rocprofv3 --kernel-trace --stats (tools/r04_emit_tiles.sh).  k_emit_points<KIND, RGB>: the colourless form no longer issues the three",
         "# masked colour loads per point, and a tile without a match returns before its first load (r04_emit_tiles.log).", ""]
    for q in ("ca13_XL", "ca13_S"):
        L.append(f"== {q} ==")
        L += ["  " + l.strip() for l in open(f"{O}/emit_probe_{q}.log") if "matches" in l]
        L += kstats(f"{O}/prof_emit_{q}/e_kernel_stats.csv")
        L.append("")
    open(f"{P}/r04_emit_probe.log", "w").write("\n".join(L))

for src, dst in (("bench_n1.json", "r04_bench_n1.json"), ("bench_torchrun_n1.json", "r04_bench_torchrun_n1.json"), ("bench_profiled.json", "r04_bench_n1_profiled_run.json"),
                 ("prof_bench/b_kernel_stats.csv", "r04_bench_n1_kernel_stats.csv"), ("prof_bench/b_kernel_trace.csv", "r04_bench_n1_kernel_trace.csv"),
                 ("collector_timings.log", "r04_collector_timings.log"), ("cli_e2e.log", "r04_cli_e2e.log")):
    if os.path.exists(f"{O}/{src}"):
        shutil.copy(f"{O}/{src}", f"{P}/{dst}")
if os.path.exists(f"{O}/pmc_latest.json"):
    pmc = json.load(open(f"{O}/pmc_latest.json"))
    pmc["git_head"] = head
    for dst in ("pmc_latest.json", "r04_pmc_traffic.json"):
        json.dump(pmc, open(f"{P}/{dst}", "w"), indent=1)
if os.path.exists(f"{O}/experiments/product.txt"):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "manual", "format_experiments.py"), f"{O}/experiments"], capture_output=True, text=True)
    if r.returncode == 0:
        open(f"{P}/r04_query_experiments.txt", "w").write(r.stdout)

if os.path.exists(f"{O}/emit_tiles.log"):
    L = [f"# round 4 (git {head}): buffer collector on one 163 M-point ca13 file, boxes that keep 100 % / 10 % / 1 % of the file's x range (FRAC), the file in",
         "# generator order (SORTED=0: every 2048-point tile holds a few matches) and sorted along x (SORTED=1: the matches are one run of the file, the",
         "# emit skips every other tile); tools/emit_probe.py through tools/r04_emit_tiles.sh, four writer settings each:",
         "#   EMIT_PARK_MAX=256 EMIT_SPARSE_MAX=64  shipped: a tile with at most 256 matches leaves them as 16-byte words in the count pass (k_emit_parked)",
         "#   EMIT_PARK_MAX=0   EMIT_SPARSE_MAX=64  the round's first step: one wave per tile with at most 64 matches, from the match bits (k_emit_sparse)",
         "#   EMIT_PARK_MAX=0   EMIT_SPARSE_MAX=0   round 3's kernels",
         "#   EMIT_PARK_MAX=0   EMIT_SPARSE_MAX=256 the one-wave writer up to 256 matches (slower at 10 % kept: dropped)",
         "# The kernel lines at the end: SORTED=1 FRAC=0.1 under rocprofv3 --kernel-trace --stats.  GB/s = algorithmic bytes (13 B read per point + 31 B",
         "# written per match) / wall time.", ""]
    L += [l.rstrip() for l in open(f"{O}/emit_tiles.log") if "amdgpu.ids" not in l]
    open(f"{P}/r04_emit_tiles.log", "w").write("\n".join(L) + "\n")

if os.path.exists(f"{O}/final_stamps.txt"):
    L = [f"# round 4 (final state, git {head}): cycles per stage and wave of the two folds, from the timing build (make -C csrc stamps -> libpcq_stamps.so,",
         "# PCQ_LAB=stamps PCQ_TIMING=1 tools/grid_probe.py; tools/r04_final_grid.sh).  The stamps cost a wait for everything in flight at some stages, so",
         "# the build is ~15 % slower than the product and shifts time towards the stages that wait; shares, not times.  Stages:",
         "#   k_fold_stream (100 m): [0] table clear + earlier winners [1] first batches [2] issue: chunk addresses + loads [4] decode, cell, key, distance",
         "#     [5] hash, probe, compare, lower the minimum [7] survivor append [8] hand-over (the next chunk has arrived) [9] end-of-stream barrier",
         "#     [3] exact pass over the survivors [6] places of the cells [12] winners' records gathered [11] keys and records issued [13] ... and taken",
         "#     [10] last barrier + thread 0's bookkeeping",
         "#   k_fold_dense (10 m): [9] rotating registers [0] ranges, asking for the next partition [1] decode, cell, key, distance [2] barrier: table clean",
         "#     [3] phase 1: inserts, minimum distance [4] cell count + barrier [5] phase 2 + ranks + barrier [6] ranks [7] next partition's tuples arrived",
         "#     [8] winners' stores, slots reset", ""]
    L += [l.rstrip() for l in open(f"{O}/final_stamps.txt")]
    open(f"{P}/r04_grid_stamps.txt", "w").write("\n".join(L) + "\n")
if os.path.exists(f"{O}/rocprof_query.log"):
    L = [f"# round 4 (git {head}): the `query` binary under rocprofv3 with kernel and memory-copy tracing, ONE run under a hard limit of 150 s",
         "# (tests/manual/rocprof_query.sh; 4 ca13 files of 4 M points, bounds XL + --density 10).  Round 3 saw this command print its answer, write its",
         "# traces and not end within 200 s; since the contexts are released by the workers and by main() before it returns (not by destructors of",
         "# thread-local objects at thread exit) it ends by itself: exit code and wall time in the last line.", ""]
    L += [l.rstrip().replace("/tmp/code/igd-geo__adhoc-queries-pointclouds/repo", "$REPO") for l in open(f"{O}/rocprof_query.log") if "pool block" not in l]
    open(f"{P}/r04_rocprof_query.log", "w").write("\n".join(L) + "\n")

# the K1 line, from the committed summary
ks = f"{P}/r04_bench_n1_kernel_stats.csv"
if os.path.exists(ks):
    for r in csv.DictReader(open(ks)):
        if "k_bounds_count_batch_pipe" in r["Name"]:
            avg, mn, calls = float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, int(r["Calls"])
            gb = 12 * 2_608_000_000 / 1e9
            print(f"K1 under rocprofv3 ({os.path.basename(ks)}): {calls} launches, average {avg:.3f} ms, min {mn:.3f} ms -> "
                  f"{gb / (avg * 1e-3):.0f} GB/s = {gb / (avg * 1e-3) / 8000:.3f} of 8 TB/s (min: {gb / (mn * 1e-3) / 8000:.3f})")
print("profiles written for", head)
