#!/bin/bash
# The `query` binary under rocprofv3 with kernel and memory-copy tracing: round 3 saw this command print its answer, write its
# traces and not end within 200 s.  ONE run, under a hard timeout, command and log kept -> profiles/r04_rocprof_query.log
# usage (on the GPU box): bash tests/manual/rocprof_query.sh   (under tests/: the files are written by the oracle's generator)
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
cd $GRAFT_REPO_ROOT
D=$(mktemp -d /tmp/pcq_prof_XXXX)
python3 - "$D" <<'PY'
import os, sys, importlib
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import _oracle
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
o = _oracle.Oracle()
for i, s in enumerate(specs.synth_ca13(points_per_file=4_000_000, files=4)):
    o.synth_write(s, os.path.join(sys.argv[1], f"t{i}.last"), threads=8)
PY
cd /tmp && export TMPDIR=/tmp
XL="643431.76;3883547.565;-46194.145;736910.93;3977026.735;47285.025"
CMD="rocprofv3 --kernel-trace --memory-copy-trace -d $O/prof_query -o q --output-format csv -- $GRAFT_REPO_ROOT/adhoc-queries-pointclouds_amd/host/query -i $D --optimized --parallel --bounds $XL --density 10"
{
  echo "# $CMD"
  echo "# (PCQ_TIMING=1; hard limit 150 s)"
  START=$(date +%s%N)
  PCQ_TIMING=1 timeout -k 10 150 $CMD
  RC=$?
  END=$(date +%s%N)
  echo "# exit code $RC after $(( (END - START) / 1000000 )) ms"
  ls $O/prof_query/*/ 2>/dev/null | head -20
} > $O/rocprof_query.log 2>&1
tail -30 $O/rocprof_query.log
rm -rf "$D"
