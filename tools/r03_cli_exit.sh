# What the end of a `query` process costs: wall clock of the CLI with the normal return path (PCQ_EXIT=full) against _exit after the
# flush with nothing released (PCQ_EXIT=fast, the default outside profilers), 12 runs each, alternating; small input so that start-up
# and exit dominate -> profiles/r03_cli_exit.log.  (An earlier state also had "release the contexts, then _exit": 13.6 ms outside
# main() like fast, plus the 9 ms of the releases inside.)
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, subprocess, tempfile, time, statistics
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import importlib, _oracle
specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
o = _oracle.Oracle()
d = tempfile.mkdtemp(prefix="pcq_exit_", dir="/tmp")
for i, s in enumerate(specs.synth_ca13(points_per_file=2_000_000, files=4)):
    o.synth_write(s, os.path.join(d, f"t{i}.last"), threads=8)
q = "adhoc-queries-pointclouds_amd/host/query"
xl = "643431.76;3883547.565;-46194.145;736910.93;3977026.735;47285.025"
for name, args in (("count", ["--bounds", xl]), ("count, sequential driver", ["--bounds", xl, "SEQ"]), ("density 100", ["--bounds", xl, "--density", "100"]), ("header only (no GPU)", ["--bounds", "0;0;0;1;1;1"])):
    times = {"full": [], "fast": []}
    inproc = {"full": [], "fast": []}
    for rep in range(12):
        for mode in ("full", "fast"):
            env = dict(os.environ, PCQ_TIMING="1")
            env["PCQ_EXIT"] = mode
            t0 = time.perf_counter()
            a = [x for x in args if x != "SEQ"] + ([] if "SEQ" in args else ["--parallel"])
            r = subprocess.run([q, "-i", d, "--optimized"] + a, capture_output=True, text=True, env=env)
            dt = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr
            times[mode].append(dt * 1e3)
            ip = [l for l in r.stderr.splitlines() if "total in-process" in l]
            inproc[mode].append(float(ip[0].split("in-process")[1].split("ms")[0]) if ip else 0.0)
    for mode in ("full", "fast"):
        t = sorted(times[mode]); ip = sorted(inproc[mode])
        print(f"{name:26s} { {'full': 'return from main', 'fast': '_exit, nothing released'}[mode]:24s} wall min {t[0]:6.1f} median {statistics.median(t):6.1f} max {t[-1]:6.1f} ms   "
              f"in-process median {statistics.median(ip):6.1f} ms   outside main() median {statistics.median([a - b for a, b in zip(times[mode], inproc[mode])]):6.1f} ms", flush=True)
PY
