#!/bin/bash
# Randomised GPU parity tests under other seed bases (tests/test_gpu_random.py, tests/test_gpu_grid_tiles.py read PCQ_TEST_SEED_BASE).
# usage (on the GPU box): bash tools/r04_soak.sh FIRST LAST
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
cd $GRAFT_REPO_ROOT
: > $O/soak.log
for b in $(seq ${1:-1} ${2:-20}); do
  PCQ_TEST_SEED_BASE=$((b * 1000)) timeout -k 10 300 python -m pytest tests/test_gpu_random.py tests/test_gpu_grid_tiles.py -x -q > $O/soak_$b.log 2>&1
  rc=$?
  echo "seed base $((b * 1000)): rc $rc $(tail -1 $O/soak_$b.log)" >> $O/soak.log
  if [ $rc -ne 0 ]; then tail -30 $O/soak_$b.log; break; fi
  rm -f $O/soak_$b.log
done
cat $O/soak.log
