# What RCCL costs a `query` process: PCQ_TIMING lines of the CLI on four small files, -> profiles/r03_rccl_cost.log
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, subprocess, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import importlib, _oracle
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
o = _oracle.Oracle()
d = tempfile.mkdtemp(prefix="pcq_rccl_", dir="/tmp")
for i, s in enumerate(specs.synth_ca13(points_per_file=2_000_000, files=4)):
    o.synth_write(s, os.path.join(d, f"t{i}.last"), threads=8)
q = "adhoc-queries-pointclouds_amd/host/query"
xl = "643431.76;3883547.565;-46194.145;736910.93;3977026.735;47285.025"
# one GPU: the default merge (a device-to-device copy), then the all-reduce forced through the real RCCL calls at one rank
# two device slots on the one GPU (PCQ_TEST_DEVICE_SLOTS): the N > 1 paths — host sum; then PCQ_MERGE=rccl, whose helper thread loads
# librccl while the workers start (their contexts and first files wait for it) and whose communicator RCCL refuses (repeated device)
for env in ({}, {"PCQ_MERGE": "rccl"}, {"PCQ_TEST_ALLREDUCE_FAIL": "late"}, {"PCQ_TEST_DEVICE_SLOTS": "0,0"}, {"PCQ_TEST_DEVICE_SLOTS": "0,0", "PCQ_MERGE": "rccl"}):
    for rep in range(2):
        r = subprocess.run([q, "-i", d, "--optimized", "--parallel", "--bounds", xl], capture_output=True, text=True, env=dict(os.environ, PCQ_TIMING="1", **env))
        print(env, [l for l in r.stderr.splitlines() if "RCCL" in l or "rccl" in l or "count merge" in l or "all-reduce" in l or "merged" in l or "workers done" in l or "context on" in l or "searched in" in l or "warning" in l], r.stdout.splitlines()[1:2])
PY
