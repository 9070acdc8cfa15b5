"""Reads rocprofv3 kernel-trace + memory-copy-trace CSVs and reports how much of the H2D copy time runs while a kernel
of the scan is executing.  usage: overlap_report.py DIR"""
import csv, glob, sys
d = sys.argv[1]
kern, cop = [], []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "HOST_TO_DEVICE" in r.get("Direction", "") or "H2D" in r.get("Direction", ""):
            cop.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(float(r.get("Bytes", 0) or 0)) if "Bytes" in r else 0))
kern.sort(); cop.sort()
big = [c for c in cop if c[1] - c[0] > 50_000]  # the staged chunks (>= 50 us), not the small uploads
def busy_inside(a, b):
    t = 0
    for s, e, _ in kern:
        if e <= a: continue
        if s >= b: break
        t += min(e, b) - max(s, a)
    return t
tot = sum(e - s for s, e, _ in big)
ov = sum(busy_inside(s, e) for s, e, _ in big)
names = {}
for s, e, n in kern:
    n = n.replace("(anonymous namespace)::", "").split("(")[0]
    names[n] = names.get(n, 0) + (e - s)
print(f"H2D copies of staged chunks: {len(big)}, {tot / 1e6:.2f} ms in all; kernels were executing during {ov / 1e6:.2f} ms of that")
print("kernel time by name (ms):", {k: round(v / 1e6, 2) for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:8]})
if big:
    span = big[-1][1] - big[0][0]
    print(f"span first copy start -> last copy end: {span / 1e6:.2f} ms; copy engine busy {100 * tot / span:.0f} % of it")

# timeline of each scan (copies separated by less than 3 ms belong to one scan): when the copy engine and the kernels ran
groups, cur = [], []
for c in big:
    if cur and c[0] - cur[-1][1] > 3_000_000:
        groups.append(cur); cur = []
    cur.append(c)
if cur: groups.append(cur)
for gi, g in enumerate(groups):
    a, b = g[0][0], g[-1][1]
    ks = [(s, e, n) for s, e, n in kern if e > a and s < b + 3_000_000]
    inside = sum(min(e, b) - max(s, a) for s, e, n in ks if s < b and e > a)
    after = sum(e - max(s, b) for s, e, n in ks if e > b)
    print(f"scan {gi}: {len(g)} chunk copies over {(b - a) / 1e6:.2f} ms (engine busy {sum(e - s for s, e, _ in g) / 1e6:.2f} ms); "
          f"{len(ks)} kernels, {inside / 1e6:.2f} ms of kernel time under the copies, {after / 1e6:.2f} ms after the last copy ended")
if "--timeline" in sys.argv and groups:
    g = groups[int(sys.argv[sys.argv.index("--timeline") + 1])]
    a, b = g[0][0], g[-1][1]
    ev = [(s, e, "H2D chunk") for s, e, _ in g] + [(s, e, n.replace("(anonymous namespace)::", "").split("(")[0]) for s, e, n in kern if e > a and s < b + 3_000_000]
    for s, e, n in sorted(ev):
        print(f"  {(s - a) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {n}")
