#!/bin/bash
set -e
cd /root/repo
D=$(mktemp -d /tmp/pcq_lz_XXXX); trap 'rm -rf "$D"' EXIT
python3 tests/manual/make_experiment_datasets.py "$D" --navvis-points 56200000 --doc-points 1000 --ca13-points 1000 --formats last,lazer > /dev/null
Q=adhoc-queries-pointclouds_amd/host/query
XL="-23.108;-21.261;-10.029;28.588;27.123;5.959"
for f in last lazer; do
  for rep in 1 2 3; do
    TIMEFORMAT="process ($f): real %R s user %U sys %S"
    time (PCQ_TIMING=1 $Q -i "$D/navvis3/$f" --bounds "$XL" --optimized --parallel 2>&1 | grep -E "pcq\]|Found" | tr '\n' ' '; echo)
  done
done
