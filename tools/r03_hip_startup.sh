# HIP start-up of a fresh process, step by step, and the cost of handing pre-read bytes to the GPU -> profiles/r03_hip_startup.log
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/r03_hip_startup tools/src/r03_hip_startup.hip || exit 1
for i in 1 2 3; do echo "== process $i =="; timeout -k 10 120 /tmp/r03_hip_startup 1024 || exit 1; done
