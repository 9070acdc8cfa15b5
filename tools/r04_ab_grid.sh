#!/bin/bash
# A/B on ONE box: the grid collector of the previous build (tools/ab/old, staged by hand) against the working tree's, and
# the working tree's with padded tile blocks; wall clock of scan + fold from tools/grid_probe.py, no profiler.
# usage (on the GPU box): bash tools/r04_ab_grid.sh [CELLS]
CELLS=${1:-"100 10"}
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cell in $CELLS; do
  echo "== old build, $cell m (rep $rep)"; python3 tools/ab/old/tools/grid_probe.py ca13_XL $cell 163000000 4 2>&1 | grep cells | tail -2
  echo "== new build, $cell m (rep $rep)"; python3 tools/grid_probe.py ca13_XL $cell 163000000 4 2>&1 | grep cells | tail -2
done; done
for pad in 64 256 320 1280; do
  for cell in $CELLS; do
    echo "== new build, block pad $pad x 16 B, $cell m"; GRID_BLOCK_PAD=$pad python3 tools/grid_probe.py ca13_XL $cell 163000000 4 2>&1 | grep cells | tail -2
  done
done
echo "== new build, 24-byte tuples"; for cell in $CELLS; do GRID_TUPLE16=0 python3 tools/grid_probe.py ca13_XL $cell 163000000 4 2>&1 | grep cells | tail -2; done
echo "== new build, stream fold off"; GRID_STREAM=0 python3 tools/grid_probe.py ca13_XL 100 163000000 4 2>&1 | grep cells | tail -2
