#!/bin/bash
# quick A/B on one box: old build (tools/ab/old) against the working tree, then per-kernel times of the working tree (random order)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
for cell in 100 10; do
  echo "== old build, $cell m"; python3 tools/ab/old/tools/grid_probe.py ca13_XL $cell 163000000 4 2>&1 | grep cells | tail -2
  echo "== new build, $cell m"; python3 tools/grid_probe.py ca13_XL $cell 163000000 4 2>&1 | grep cells | tail -2
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_old_10 -o g --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/ab/old/tools/grid_probe.py ca13_XL 10 163000000 4 > $O/old_10.log 2>&1
echo "== kernels of the OLD build, 10 m"; python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/prof_old_10/g_kernel_stats.csv 0.02 | grep -v "at::\|elementwise\|synth\|vectorized"
for cell in 100 10; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_q_$cell -o g --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/grid_probe.py ca13_XL $cell 163000000 4 > $O/q_$cell.log 2>&1
  echo "== kernels, $cell m"; python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/prof_q_$cell/g_kernel_stats.csv 0.02 | grep -v "at::\|elementwise\|synth\|vectorized"
done
