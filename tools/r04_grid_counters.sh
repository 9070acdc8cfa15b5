#!/bin/bash
# SQ and HBM-traffic counters of the grid collector's kernels (XL box, one 163 M-point ca13 file; 2 scans + folds per pass).
# usage (on the GPU box): bash tools/r04_grid_counters.sh TAG "CELLS" [random|coherent]
TAG=${1:-cnt}; CELLS=${2:-"10 100"}; ORDER=${3:-random}
if [ $ORDER = coherent ]; then export COHERENT=10; fi
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cell in $CELLS; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $set -d $O/pmc_${TAG}_${cell}_$i -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/grid_probe.py ca13_XL $cell 163000000 2 > $O/pmc_${TAG}_${cell}_$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $O/pmc_${TAG}_${cell}_$i.log; exit 1; }
    echo "cell $cell pass $i done"
  done
  python3 $GRAFT_REPO_ROOT/tools/pmc_sum.py $O/pmc_${TAG}_${cell}_1 $O/pmc_${TAG}_${cell}_2 $O/pmc_${TAG}_${cell}_3 $O/pmc_${TAG}_${cell}_4 > $O/pmc_${TAG}_${ORDER}_${cell}.txt
  rm -rf $O/pmc_${TAG}_${cell}_[1-4]
done
