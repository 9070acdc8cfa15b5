#!/bin/bash
# k_level2 with and without the selector in the tuples' spare top bytes, same box, three times each (its time moved by 20 %
# between two runs of the same code)
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
for rep in 1 2; do for t16 in 1 2 0; do
  GRID_TUPLE16=$t16 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_l2_${t16}_$rep -o g --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/grid_probe.py ca13_XL 10 163000000 4 > $O/l2_${t16}_$rep.log 2>&1
  echo "== GRID_TUPLE16=$t16 rep $rep: $(grep cells $O/l2_${t16}_$rep.log | tail -1 | cut -c1-80)"; python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/prof_l2_${t16}_$rep/g_kernel_stats.csv 0.5 | grep -v "at::\|elementwise\|synth\|vectorized"
done; done
