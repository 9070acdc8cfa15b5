#!/bin/bash
# Buffer collector: parity tests, then the emit probe (XL and S boxes, one 163 M-point ca13 file) under rocprofv3.
TAG=${1:-emit}
O=$GRAFT_REPO_ROOT/gpurun_out/r02; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 420 python -m pytest tests/test_gpu_scan.py tests/test_gpu_host.py tests/test_gpu_random.py -m gpu -x -q -k "buffer or points or random or las" > $O/${TAG}_tests.log 2>&1
rc=$?; tail -3 $O/${TAG}_tests.log
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
for q in ca13_XL ca13_S; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_$q -o e --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/emit_probe.py $q 163000000 5 > $O/${TAG}_probe_$q.log 2>&1 || exit 1
  grep matches $O/${TAG}_probe_$q.log | tail -2
  python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/prof_${TAG}_$q/e_kernel_stats.csv 0.01
done
