# Fixed cost of a grid fold: tiny files (the kernels have nothing to do), scan + count wall time -> gpurun_out/r03/grid_small.log
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03; O=gpurun_out/r03/grid_small.log; : > $O
for n in 1000 100000 2000000; do for cell in 100 10; do
  timeout -k 10 120 python tools/grid_probe.py ca13_XL $cell $n 6 2>/dev/null | tail -3 >> $O || exit 1
done; done
cat $O
