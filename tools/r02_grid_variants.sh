#!/bin/bash
# Lab: the grid probe at one cell size cycling through GRID_VARIANT shapes in ONE process (same box, same buffers: box-to-box spread
# moves the partition kernels by +-15 %), kernel stats by kernel name.
# usage (on the GPU box): bash tools/r02_grid_variants.sh CELL "V,V,..." REPEATS TAG
CELL=${1:-10}; VARS=${2:-"0,1,2,3"}; REP=${3:-12}; TAG=${4:-var}
O=$GRAFT_REPO_ROOT/gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp PCQ_LAB=1
GRID_VARIANT=$VARS timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_${CELL} -o g --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/grid_probe.py ca13_XL $CELL 163000000 $REP > $O/${TAG}_${CELL}.log 2>&1 || { tail -5 $O/${TAG}_${CELL}.log; exit 1; }
grep cells $O/${TAG}_${CELL}.log | tail -$REP
python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/prof_${TAG}_${CELL}/g_kernel_stats.csv 0.3
