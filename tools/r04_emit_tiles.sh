# Buffer collector on one 163 M-point file: boxes that keep 100 % / 50 % / 10 % / 1 % of the file, generator order and x-sorted order
# (a tile of 2048 points without a match is not read a second time) -> profiles/r04_emit_tiles.log
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
O=gpurun_out/r04/emit_tiles.log; : > $O
# writers: "park sparse" = emit_park_max emit_sparse_max — 256 64 shipped; 0 64 = one wave per thin tile from the match bits (the round's first
# step); 0 0 = round 3's kernels; 0 256 = the one-wave writer up to 256 matches
for sorted in "" 1; do for frac in 1.0 0.1 0.01; do for w in "256 64" "0 64" "0 0" "0 256"; do
  set -- $w
  echo "== SORTED=${sorted:-0} FRAC=$frac EMIT_PARK_MAX=$1 EMIT_SPARSE_MAX=$2 ==" >> $O
  EMIT_PARK_MAX=$1 EMIT_SPARSE_MAX=$2 SORTED=$sorted FRAC=$frac timeout -k 10 300 python tools/emit_probe.py ca13_XL 163000000 4 >> $O 2>&1 || exit 1
done; done; done
cd /tmp && export TMPDIR=/tmp
for q in ca13_XL ca13_S; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04/prof_emit_$q -o e --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/emit_probe.py $q 163000000 5 > $GRAFT_REPO_ROOT/gpurun_out/r04/emit_probe_$q.log 2>&1 || exit 1
done
SORTED=1 FRAC=0.1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04/prof_emit_tiles -o e --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/emit_probe.py ca13_XL 163000000 4 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/kstats.py $(find gpurun_out/r04/prof_emit_tiles -name 'e_kernel_stats.csv' | head -1) >> $O 2>&1
cat $O
