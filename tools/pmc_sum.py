"""Sums rocprofv3 --pmc counter_collection CSVs per kernel and counter.  usage: pmc_sum.py DIR [DIR ...]
Prints, per kernel (dispatches averaged), every counter found; FETCH_SIZE / WRITE_SIZE are left in the profiler's units
(KB; FETCH_SIZE needs the x2 gfx950 correction, see MI355X_MICROARCH.md)."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[n].add(r["Dispatch_Id"])
    print("==", d)
    for n, cs in acc.items():
        k = len(calls[n])
        if max(cs.values()) / k < 1e5:
            continue
        print(f"{n[:70]} ({k} dispatches, per dispatch)")
        print("    " + "  ".join(f"{c}={v / k:.3g}" for c, v in sorted(cs.items())))
