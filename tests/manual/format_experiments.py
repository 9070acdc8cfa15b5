"""Manual: merges the product.txt / oracle.txt written by run_experiments.sh into one table.
   python tests/manual/format_experiments.py gpurun_out/experiments > profiles/rNN_query_experiments.txt"""
import os
import sys

d = sys.argv[1]


def read(name):
    rows = {}
    for line in open(os.path.join(d, name)):
        f = line.strip().split(";")
        if len(f) == 4:
            rows[f[0]] = tuple(float(x) for x in f[1:])
    return rows


prod, orac = read("product.txt"), read("oracle.txt")
print("# run_query_experiments (paper protocol: one `query --optimized --parallel` process per run, wall clock of the")
print("# whole process incl. HIP start-up, 5 runs, warm page cache; one second in front of every run where the reference has `sync; purge`")
print("# — started right behind another GPU process a query waits 0.1-0.2 s in hsa_init: r03_query_experiments_back_to_back.txt is that case) on")
print("# scaled-down synthetic datasets:")
print("# " + " | ".join(l.strip() for l in open(os.path.join(d, "datasets.txt")) if l.strip()))
print("# product = this repository's CLI on one MI355X; oracle = single-threaded C restatement of the reference (1 run)")
print(f"{'experiment':<30} {'product mean':>12} {'median':>8} {'stddev':>8} {'oracle':>10} {'oracle/product':>14}")
for name, (mean, median, sd) in prod.items():
    o = orac.get(name)
    print(f"{name:<30} {mean:12.3f} {median:8.3f} {sd:8.3f} " + (f"{o[0]:10.3f} {o[0] / mean:14.1f}" if o else f"{'-':>10} {'-':>14}"))
