# HIP start-up of a fresh process under a few runtime settings: 10 processes each, medians of hipInit and of the first two queues
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/r03_hip_startup tools/src/r03_hip_startup.hip || exit 1
run() {  # name, env assignments...
  name=$1; shift
  for i in 1 2 3 4 5 6 7 8 9 10; do env "$@" timeout -k 10 60 /tmp/r03_hip_startup 16 | head -3 | awk '{print $(NF-1)}' | tr '\n' ' '; echo; done > /tmp/su_$name.txt
  python3 - "$name" <<'PY'
import sys, statistics
rows = [[float(x) for x in l.split()] for l in open(f"/tmp/su_{sys.argv[1]}.txt") if len(l.split()) == 3]
init = [r[0] for r in rows]; q = [r[2] for r in rows]; tot = [sum(r) for r in rows]
print(f"{sys.argv[1]:28s} hipInit median {statistics.median(init):6.1f} (min {min(init):6.1f} max {max(init):6.1f})   two queues median {statistics.median(q):6.1f} (min {min(q):6.1f} max {max(q):6.1f})   total median {statistics.median(tot):6.1f} ms", flush=True)
PY
}
# sdma_off = HSA_ENABLE_SDMA=0, max_hw_queues_2 = GPU_MAX_HW_QUEUES=2, no_interrupt = HSA_ENABLE_INTERRUPT=0
run default PCQ_X=1
run sdma_off HSA_ENABLE_SDMA=0
run max_hw_queues_2 GPU_MAX_HW_QUEUES=2
run no_interrupt HSA_ENABLE_INTERRUPT=0
run default_again PCQ_X=1
