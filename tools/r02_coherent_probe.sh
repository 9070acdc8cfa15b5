O=$GRAFT_REPO_ROOT/gpurun_out/r02; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for cell in 100 10; do
COHERENT=10 rocprofv3 --kernel-trace --stats -d $O/prof_coh_$cell -o g --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/grid_probe.py ca13_XL $cell 163000000 3 > $O/coh_$cell.log 2>&1 || { tail -5 $O/coh_$cell.log; exit 1; }
echo "== coherent $cell"; grep cells $O/coh_$cell.log | tail -2; python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/prof_coh_$cell/g_kernel_stats.csv 0.3 | grep -v "at::\|elementwise\|sort\|Sort\|radix\|Radix\|synth"
done
