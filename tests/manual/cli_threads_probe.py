"""The `query` CLI's per-file times by the number of threads that fill a staging chunk (PCQ_COPY_THREADS): bounds XL over 16 ca13 files of 20 M
points, three processes each.  Manual check under tests/ because the files are written by the oracle's generator."""
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
d = tempfile.mkdtemp(prefix="pcq_thr_", dir="/tmp")
for i, s in enumerate(specs.synth_ca13(points_per_file=20_000_000, files=16)):
    o.synth_write(s, os.path.join(d, f"tile{i:02d}.last"), threads=32)
xl = "643431.76;3883547.565;-46194.145;736910.93;3977026.735;47285.025"
for rep in range(3):
    for thr in [int(a) for a in sys.argv[1:]] or [8, 4, 12, 16]:
        time.sleep(0.5)
        r = subprocess.run([QUERY, "-i", d, "--optimized", "--parallel", "--bounds", xl], capture_output=True, text=True,
                           env=dict(os.environ, PCQ_TIMING="1", PCQ_EXIT="fast", PCQ_COPY_THREADS=str(thr)))
        files = [float(l.split(" searched in ")[1].split(" ms")[0]) for l in r.stderr.splitlines() if " searched in " in l]
        print("copy threads %2d: first file %.1f ms, median %.1f, min %.1f, sum %.1f" % (thr, files[0], sorted(files)[len(files) // 2], min(files), sum(files)), flush=True)
for f in os.listdir(d):
    os.remove(os.path.join(d, f))
os.rmdir(d)
