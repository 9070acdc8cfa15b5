"""The `query` CLI's first file under PCQ_TIMING=1: the stamps of the first scan of a context, three processes (bounds XL over 16 ca13 files
of 20 M points).  Manual check under tests/ because the files are written by the oracle's generator."""
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
d = tempfile.mkdtemp(prefix="pcq_first_", dir="/tmp")
for i, s in enumerate(specs.synth_ca13(points_per_file=20_000_000, files=16)):
    o.synth_write(s, os.path.join(d, f"tile{i:02d}.last"), threads=32)
xl = "643431.76;3883547.565;-46194.145;736910.93;3977026.735;47285.025"
for rep in range(6):
    time.sleep(1.0)
    mode = "2"  # (the default: in place while the copy path is set up; 0 = never, 1 = always: see profiles/r04_cli_first_file.log)
    r = subprocess.run([QUERY, "-i", d, "--optimized", "--parallel", "--bounds", xl], capture_output=True, text=True,
                       env=dict(os.environ, PCQ_TIMING="1", PCQ_EXIT="fast", PCQ_HOST_IN_PLACE=mode))
    print("PCQ_HOST_IN_PLACE=" + mode)
    lines = [l for l in r.stderr.splitlines() if "first scan" in l or "staging" in l or "ready after" in l]
    files = [float(l.split(" searched in ")[1].split(" ms")[0]) for l in r.stderr.splitlines() if " searched in " in l]
    print("\n".join(lines))
    print("first file %.1f ms, median %.1f, sum %.1f\n" % (files[0], sorted(files)[len(files) // 2], sum(files)), flush=True)
for f in os.listdir(d):
    os.remove(os.path.join(d, f))
os.rmdir(d)
