#!/bin/bash
# Does the scatter's process-to-process spread follow where the pool's blocks land?  Four processes, each: the blocks' addresses
# (PCQ_TIMING=1) and the kernel stats of the grid probe.
O=$GRAFT_REPO_ROOT/gpurun_out/r02; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PCQ_TIMING=1
for k in 1 2 3 4; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_place_$k -o g --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/grid_probe.py ca13_XL 10 163000000 3 > $O/place_$k.log 2>&1 || exit 1
  echo "== process $k"; grep "pool block" $O/place_$k.log | head -8; python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/prof_place_$k/g_kernel_stats.csv 1.0 | grep -v synth
done
