#!/bin/bash
# Round-2 measurement set B: grid collector kernel stats + SQ / HBM-traffic counters, buffer collector, resident dataset rate,
# the paper's experiments.
O=$GRAFT_REPO_ROOT/gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cell in 100 10; do
  rocprofv3 --kernel-trace --stats -d $O/prof_grid_$cell -o g --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/grid_probe.py ca13_XL $cell 163000000 4 > $O/grid_probe_$cell.log 2>&1; echo "grid $cell rc $?"; grep cells $O/grid_probe_$cell.log | tail -2
done
bash $GRAFT_REPO_ROOT/tools/r02_grid_counters.sh final "100 10" || exit 1
for q in ca13_XL ca13_S; do
  rocprofv3 --kernel-trace --stats -d $O/prof_emit_$q -o e --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/emit_probe.py $q 163000000 5 > $O/emit_probe_$q.log 2>&1; echo "emit $q rc $?"; grep matches $O/emit_probe_$q.log | tail -1
done
rocprofv3 --kernel-trace --stats -d $O/prof_resident -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/resident_rate.py > $O/resident_rate.log 2>&1; echo "resident rc $?"; tail -3 $O/resident_rate.log
python3 $GRAFT_REPO_ROOT/tools/copy_ceiling.py > $O/copy_ceiling.log 2>&1; tail -4 $O/copy_ceiling.log
cd $GRAFT_REPO_ROOT && bash tests/manual/run_experiments.sh $O/experiments > $O/experiments.log 2>&1; echo "experiments rc $?"; tail -3 $O/experiments.log
