#!/bin/bash
# What kind of vector instructions the grid collector's kernels issue (the f64 ones run at half rate on CDNA4): SQ_INSTS_VALU_* by
# type, one --pmc pass of eight counters per group, no tracing domains.  usage (GPU box): bash tools/r03_grid_valu_mix.sh "100 10"
CELLS=${1:-"100 10"}
O=$GRAFT_REPO_ROOT/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*\|SQ_ACTIVE_INST[A-Z0-9_]*\|SQ_INST_CYCLES[A-Z0-9_]*\|SQ_VALU_MFMA[A-Z0-9_]*" | sort -u > $O/valu_counters_avail.txt
cat $O/valu_counters_avail.txt | tr '\n' ' '; echo
for cell in $CELLS; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
             "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $set -d $O/mix_${cell}_$i -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/grid_probe.py ca13_XL $cell 163000000 2 > $O/mix_${cell}_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/mix_${cell}_$i.log; exit 1; }
  done
  python3 $GRAFT_REPO_ROOT/tools/pmc_sum.py $O/mix_${cell}_1 $O/mix_${cell}_2 > $O/valu_mix_$cell.txt
  rm -rf $O/mix_${cell}_[12]
  cat $O/valu_mix_$cell.txt
done
