O=$GRAFT_REPO_ROOT/gpurun_out/r02; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PCQ_LAB=1
GRID_VARIANT=0,4096,8192 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_seqw -o g --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/grid_probe.py ca13_XL 100 163000000 9 > $O/seqw.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/prof_seqw/g_kernel_stats.csv 0.3
