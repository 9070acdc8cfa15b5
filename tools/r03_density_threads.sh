# `query --density`: one host thread per GPU against two (a file's fold ends on a synchronisation, which drains a single thread's
# pipeline; a second context costs its own start-up).  8 runs each, one second apart -> profiles/r03_density_threads.log
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, subprocess, tempfile, time, statistics
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import importlib, _oracle
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
o = _oracle.Oracle()
q = "adhoc-queries-pointclouds_amd/host/query"
xl = "643431.76;3883547.565;-46194.145;736910.93;3977026.735;47285.025"
for files, pts in ((16, 20_000_000), (4, 20_000_000), (64, 2_000_000)):
    d = tempfile.mkdtemp(prefix="pcq_dt_", dir="/tmp")
    ss = specs.synth_ca13(points_per_file=pts, files=16)
    for i in range(files):
        o.synth_write(ss[i % 16], os.path.join(d, f"t{i:02d}.last"), threads=16)
    for name, args in (("density 100", ["--density", "100"]), ("density 10", ["--density", "10"]), ("count", [])):
        for tpg in (1, 2, 0):
            ts = []
            for rep in range(8):
                time.sleep(1.0)
                t0 = time.perf_counter()
                r = subprocess.run([q, "-i", d, "--optimized", "--parallel", "--bounds", xl] + (["--threads-per-gpu", str(tpg)] if tpg else []) + args, capture_output=True, text=True, env=dict(os.environ, PCQ_EXIT="fast"))
                ts.append((time.perf_counter() - t0) * 1e3)
                assert r.returncode == 0, r.stderr
            ts.sort()
            print(f"{files:3d} files x {pts // 1000000:2d} M points  {name:12s} threads per GPU {tpg if tpg else 'default'}: min {ts[0]:6.1f}  median {statistics.median(ts):6.1f}  max {ts[-1]:6.1f} ms", flush=True)
    for f in os.listdir(d): os.remove(os.path.join(d, f))
    os.rmdir(d)
PY
