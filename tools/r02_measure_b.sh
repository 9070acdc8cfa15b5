#!/bin/bash
# Round-2 measurement set B: grid collector kernel stats + HBM traffic counters, resident dataset rate, the paper's experiments.
O=$GRAFT_REPO_ROOT/gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cell in 100 10; do
  rocprofv3 --kernel-trace --stats -d $O/prof_grid_$cell -o g --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/grid_probe.py ca13_XL $cell 163000000 4 > $O/grid_probe_$cell.log 2>&1; echo "grid $cell rc $?"; grep cells $O/grid_probe_$cell.log | tail -2
  rocprofv3 --pmc FETCH_SIZE -d $O/pmc_grid_fetch_$cell -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/grid_probe.py ca13_XL $cell 163000000 2 > /dev/null 2>&1; echo "fetch $cell rc $?"
  rocprofv3 --pmc WRITE_SIZE -d $O/pmc_grid_write_$cell -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/grid_probe.py ca13_XL $cell 163000000 2 > /dev/null 2>&1; echo "write $cell rc $?"
done
rocprofv3 --kernel-trace --stats -d $O/prof_resident -o r --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/resident_rate.py > $O/resident_rate.log 2>&1; echo "resident rc $?"; tail -3 $O/resident_rate.log
cd $GRAFT_REPO_ROOT && bash tests/manual/run_experiments.sh $O/experiments > $O/experiments.log 2>&1; echo "experiments rc $?"; tail -3 $O/experiments.log
