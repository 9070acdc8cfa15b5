#!/bin/bash
# Developer experiment (run on the GPU box, in its scratch copy of the tree): rebuilds libpcq.so with different
# cache-policy bits on the streaming loads of the pipelined count kernels and prints the bench roofline for each.
#   tools/load_hint_sweep.sh            (restores the committed source and library at the end)
set -e
cd "$(dirname "$0")/.."
SRC=adhoc-queries-pointclouds_amd/csrc/scan_count.hip
cp $SRC /tmp/scan_count.hip.orig
trap 'cp /tmp/scan_count.hip.orig '"$SRC"'; make -s -j8 -C adhoc-queries-pointclouds_amd/csrc all >/dev/null 2>&1' EXIT
for hint in "nt" "sc1 nt" "sc0 sc1 nt" "sc1" "sc0" ""; do
  python3 - "$hint" <<'PY'
import re, sys
hint = sys.argv[1]
p = "adhoc-queries-pointclouds_amd/csrc/scan_count.hip"
s = open("/tmp/scan_count.hip.orig").read()
out = []
for line in s.splitlines(keepends=True):
    if "global_load_dwordx4" in line and "asm" in line or line.lstrip().startswith('"global_load_dwordx4'):
        line = re.sub(r" nt(?=\\n|\")", (" " + hint) if hint else "", line)
    out.append(line)
open(p, "w").write("".join(out))
PY
  make -s -j8 -C adhoc-queries-pointclouds_amd/csrc all >/dev/null 2>&1
  for rep in 1 2; do
    python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('hint=%-12r %8.1f GB/s  frac %.4f  %.4f ms/step' % (sys.argv[1], r['achieved'], r['frac'], d['ms_per_step']), flush=True)" "$hint"
  done
done
