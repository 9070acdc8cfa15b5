O=$GRAFT_REPO_ROOT/gpurun_out/r02; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for n in 20000000 40000000 80000000 163000000; do
rocprofv3 --kernel-trace --stats -d $O/prof_n_$n -o g --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/grid_probe.py ca13_XL 10 $n 4 > $O/n_$n.log 2>&1 || exit 1
echo "== $n"; grep cells $O/n_$n.log | tail -1; python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/prof_n_$n/g_kernel_stats.csv 0.05 | grep -v synth
done
cd $GRAFT_REPO_ROOT && python -m pytest tests -m gpu -x -q > $O/pytest_full2.log 2>&1; tail -3 $O/pytest_full2.log
