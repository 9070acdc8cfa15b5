#!/bin/bash
# Round 4's closing measurements of the grid collector on one box (one 163 M-point ca13 file, XL box):
#   kernel stats at 100 m and 10 m in generator and scan-strip order      -> gpurun_out/r04/final_kernel_stats.txt
#   SQ + HBM-traffic counters, both orders                                  -> gpurun_out/r04/pmc_fin_<order>_<cell>.txt
#   per-stage cycle stamps of the two folds (the timing build)              -> gpurun_out/r04/final_stamps.txt
# usage (on the GPU box): bash tools/r04_final_grid.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
cd $GRAFT_REPO_ROOT
bash tools/r04_measure_grid.sh fin 4 > $O/final_kernel_stats.txt 2>&1 || { tail -5 $O/final_kernel_stats.txt; exit 1; }
bash tools/r04_grid_counters.sh fin "100 10" random > $O/final_counters_random.log 2>&1 || { tail -5 $O/final_counters_random.log; exit 1; }
bash tools/r04_grid_counters.sh fin "100 10" coherent > $O/final_counters_coherent.log 2>&1 || { tail -5 $O/final_counters_coherent.log; exit 1; }
{
  for order in random coherent; do for cell in 100 10; do
    if [ $order = coherent ]; then export COHERENT=10; else unset COHERENT; fi
    echo "== $order order, $cell m (libpcq_stamps.so)"
    PCQ_LAB=stamps PCQ_TIMING=1 timeout -k 10 300 python3 tools/grid_probe.py ca13_XL $cell 163000000 3 2>&1 | grep -E "stamps|cells" | tail -4
  done; done
} > $O/final_stamps.txt 2>&1
{
  echo "# the same four cases WITHOUT the profiler (tools/grid_probe.py alone; the kernel trace costs the fold's wall clock 0.2-0.4 ms)"
  for order in random coherent; do for cell in 100 10; do
    if [ $order = coherent ]; then export COHERENT=10; else unset COHERENT; fi
    echo "== $order order, $cell m"
    timeout -k 10 300 python3 tools/grid_probe.py ca13_XL $cell 163000000 5 2>&1 | grep cells | tail -4
  done; done
} > $O/final_unprofiled.txt 2>&1
grep -v "rocprim\|copyBuffer" $O/final_kernel_stats.txt
cat $O/final_unprofiled.txt
