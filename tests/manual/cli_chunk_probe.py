"""The `query` CLI's per-file times (PCQ_TIMING) under different staging chunk sizes: bounds XL over 16 ca13 files of 20 M points.
Manual check under tests/ because the files are written by the oracle's generator.  usage: cli_chunk_probe.py [CHUNK_POINTS ...]"""
import importlib
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle  # noqa: E402  (file generator)

specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
QUERY = os.path.join(ROOT, "adhoc-queries-pointclouds_amd", "host", "query")
o = _oracle.Oracle()
d = tempfile.mkdtemp(prefix="pcq_chunk_", dir="/tmp")
for i, s in enumerate(specs.synth_ca13(points_per_file=20_000_000, files=16)):
    o.synth_write(s, os.path.join(d, f"tile{i:02d}.last"), threads=32)
xl = "643431.76;3883547.565;-46194.145;736910.93;3977026.735;47285.025"
for chunk in [int(a) for a in sys.argv[1:]] or [2 << 20, 1 << 20, 1 << 19]:
    for rep in range(3):
        time.sleep(1.0)
        t0 = time.perf_counter()
        r = subprocess.run([QUERY, "-i", d, "--optimized", "--parallel", "--bounds", xl], capture_output=True, text=True,
                           env=dict(os.environ, PCQ_TIMING="1", PCQ_EXIT="fast", PCQ_CHUNK_POINTS=str(chunk)))
        dt = time.perf_counter() - t0
        files = [float(l.split(" searched in ")[1].split(" ms")[0]) for l in r.stderr.splitlines() if " searched in " in l]
        ready = [l for l in r.stderr.splitlines() if "ready after" in l or "total in-process" in l]
        print("chunk %8d points: wall %.3f s, first file %.1f ms, median %.1f, sum %.1f ms | %s" % (
            chunk, dt, files[0], sorted(files)[len(files) // 2], sum(files), "; ".join(x.replace("[pcq] ", "") for x in ready)), flush=True)
for f in os.listdir(d):
    os.remove(os.path.join(d, f))
os.rmdir(d)
