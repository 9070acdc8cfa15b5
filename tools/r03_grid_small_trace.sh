# HIP API trace of grid folds on a tiny file: which calls make up the fixed cost
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03; mkdir -p $O
timeout -k 10 200 rocprofv3 --hip-trace --kernel-trace --stats -d $O/prof_small -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/grid_probe.py ca13_XL 100 100000 6 > $O/small_trace.log 2>&1
cd $GRAFT_REPO_ROOT && python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r03/prof_small/**/s_hip_api_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last fold: from the last k_p0 launch on
idx = [i for i, r in enumerate(rows) if r["Function"] == "hipLaunchKernel" or r["Function"] == "hipModuleLaunchKernel" or "Launch" in r["Function"]]
t_end = int(rows[-1]["End_Timestamp"])
# print the last 60 calls with durations and gaps
tail = rows[-90:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {r["Function"]}')
PY
