#!/bin/bash
# Grid collector: kernel stats at 100 m and 10 m cells (one 163 M-point ca13 file, XL box), generator (random) order and
# scan-strip order (COHERENT=10, the ordered case real flight-line tiles are).
# usage (on the GPU box): bash tools/r04_measure_grid.sh TAG [REPEATS] [GRID_AGG list]
TAG=${1:-grid}; REP=${2:-4}; AGG=${3:-}
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for order in random coherent; do
for cell in 100 10; do
  if [ $order = coherent ]; then export COHERENT=10; else unset COHERENT; fi
  GRID_AGG=$AGG timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_${TAG}_${order}_$cell -o g --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/grid_probe.py ca13_XL $cell 163000000 $REP > $O/${TAG}_${order}_$cell.log 2>&1 || { tail -5 $O/${TAG}_${order}_$cell.log; exit 1; }
  echo "== $order order, $cell m"; grep cells $O/${TAG}_${order}_$cell.log | tail -$REP
  python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/prof_${TAG}_${order}_$cell/g_kernel_stats.csv 0.02 | grep -v "at::\|elementwise\|sort\|Sort\|radix\|Radix\|synth\|index_\|vectorized\|pcqgrid::k_excl\|k_winner_room\|k_scan_pi"
done; done
