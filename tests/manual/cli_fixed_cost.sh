#!/bin/bash
# developer probe: fixed per-process cost of the CLI (small dataset, so the scan itself is negligible)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
D=$(mktemp -d /tmp/pcq_fixed_XXXX)
trap 'rm -rf "$D"' EXIT
python3 "$ROOT/tests/manual/make_experiment_datasets.py" "$D" --navvis-points 1000000 --doc-points 1000000 --ca13-points 5093750 --formats last > /dev/null
Q="$ROOT/adhoc-queries-pointclouds_amd/host/query"
XL="643431.76;3883547.565;-46194.145;736910.93;3977026.735;47285.025"
S="665000;3910000;0;705000;3950000;480"
for args in "--bounds $S" "--bounds $XL" "--class 19" "--bounds $XL --density 100"; do
  for T in 1 2; do
    echo "== $args  (threads-per-gpu $T)"
    for rep in 1 2; do
      TIMEFORMAT='process: real %R s  user %U s  sys %S s'
      time (env PCQ_TIMING=1 "$Q" -i "$D/ca13/last" $args --optimized --parallel --threads-per-gpu $T 2>&1 | grep -v "file .* searched" | tail -8)
    done
  done
done
echo "== sequential XL"
env PCQ_TIMING=1 "$Q" -i "$D/ca13/last" --bounds "$XL" --optimized 2>&1 | grep -v "file .* searched" | tail -5
