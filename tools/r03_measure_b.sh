#!/bin/bash
# Round-3 measurement set B: grid collector kernel stats (generator order and scan-strip order, 100 m and 10 m), SQ / HBM-traffic
# counters of the same four cases, buffer collector, the paper's experiments.
O=$GRAFT_REPO_ROOT/gpurun_out/r03; mkdir -p $O
bash $GRAFT_REPO_ROOT/tools/r03_measure_grid.sh final 4 > $O/grid_final.log 2>&1 || { tail -5 $O/grid_final.log; exit 1; }
grep cells $O/grid_final.log | awk 'NR%4==0'
bash $GRAFT_REPO_ROOT/tools/r03_grid_counters.sh final "100 10" random || exit 1
bash $GRAFT_REPO_ROOT/tools/r03_grid_counters.sh final "100 10" coherent || exit 1
cd /tmp && export TMPDIR=/tmp
for q in ca13_XL ca13_S; do
  rocprofv3 --kernel-trace --stats -d $O/prof_emit_$q -o e --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/emit_probe.py $q 163000000 5 > $O/emit_probe_$q.log 2>&1; echo "emit $q rc $?"; grep matches $O/emit_probe_$q.log | tail -1
done
cd $GRAFT_REPO_ROOT && bash tests/manual/run_experiments.sh $O/experiments > $O/experiments.log 2>&1; echo "experiments rc $?"; tail -3 $O/experiments.log
