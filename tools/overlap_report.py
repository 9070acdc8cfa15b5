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
