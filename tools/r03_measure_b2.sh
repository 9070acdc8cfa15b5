#!/bin/bash
# Round-3 measurement set B, the grid part only (after a change to grid.hip): kernel stats and counters of the four cases, and the
# collectors' wall-clock timings.
O=$GRAFT_REPO_ROOT/gpurun_out/r03; mkdir -p $O
bash $GRAFT_REPO_ROOT/tools/r03_measure_grid.sh final 4 > $O/grid_final.log 2>&1 || { tail -5 $O/grid_final.log; exit 1; }
grep cells $O/grid_final.log | awk 'NR%4==0'
bash $GRAFT_REPO_ROOT/tools/r03_grid_counters.sh final "100 10" random || exit 1
bash $GRAFT_REPO_ROOT/tools/r03_grid_counters.sh final "100 10" coherent || exit 1
cd $GRAFT_REPO_ROOT && python tools/config_timings.py > $O/collector_timings.log 2>&1; echo "timings rc $?"
