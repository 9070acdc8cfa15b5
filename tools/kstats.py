"""Prints a rocprofv3 kernel_stats.csv compactly.  usage: kstats.py FILE [min_ms]"""
import csv, sys
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("pcqgrid::", "").replace("void ", "").split("(")[0]
    if float(r["AverageNs"]) / 1e6 >= lim:
        print(f'{n[:64]:64s} {r["Calls"]:>4s} avg {float(r["AverageNs"]) / 1e6:8.3f} ms  min {float(r["MinNs"]) / 1e6:8.3f}')
