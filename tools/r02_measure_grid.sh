#!/bin/bash
# Grid collector: parity tests, then kernel stats at 100 m and 10 m cells (one 163 M-point ca13 file, XL box).
# usage (on the GPU box): bash tools/r02_measure_grid.sh TAG
TAG=${1:-grid}
O=$GRAFT_REPO_ROOT/gpurun_out/r02; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 240 python -m pytest tests/test_gpu_scan.py tests/test_gpu_host.py -m gpu -x -q -k "grid or golden or density or alias" > $O/${TAG}_tests.log 2>&1
rc=$?; tail -3 $O/${TAG}_tests.log
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
for cell in 100 10; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_$cell -o g --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/grid_probe.py ca13_XL $cell 163000000 4 > $O/${TAG}_probe_$cell.log 2>&1 || exit 1
  grep cells $O/${TAG}_probe_$cell.log | tail -2
  cut -d, -f1-4 $O/prof_${TAG}_$cell/g_kernel_stats.csv | sed -e 's/(anonymous namespace):://g' | cut -c1-150 | head -14
done
