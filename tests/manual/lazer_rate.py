"""Manual measurement (uses the oracle's test-side LAZER writer, hence under tests/): end-to-end rate of
Searcher::search_file on a .lazer file — host LZ4 inflate + GPU scan — next to the oracle's streaming
per-point restatement of the reference on the same file.

    python tests/manual/lazer_rate.py [points] [block_size]
"""
import importlib
import json
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch  # noqa: F401  (its HIP runtime must initialise first)

torch.cuda.is_available()
import _oracle
from test_gpu_host import Q

specs = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
block = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
o = _oracle.Oracle()
q = Q()
spec = specs.synth_navvis(points_per_file=n)[0]
t0 = time.time()
image = o.synth_image(spec, transposed=True, threads=8)
lazer = o.lazer_from_last(image, block, 4, 4)
print(f"built {len(image) / 1e6:.0f} MB LAST -> {len(lazer) / 1e6:.0f} MB LAZER in {time.time() - t0:.1f} s", flush=True)
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "f.lazer")
    lazer.tofile(path)
    bmin, bmax = specs.box("navvis_L")
    out = {"points": n, "block_size": block, "lazer_MB": len(lazer) / 1e6}
    for kind in ("bounds", "class"):
        times = []
        for it in range(4):
            h = q.collector("count")
            t = time.perf_counter()
            rc = q.search_bounds(path, bmin, bmax, h)[0] if kind == "bounds" else q.search_class(path, 2, h)
            cnt = q.count(h)
            times.append(time.perf_counter() - t)
            assert rc == 0
            q.free(h)
        oc = o.count_collector()
        t = time.perf_counter()
        rc = o.search_file(path, 0 if kind == "bounds" else 1, bmin, bmax, 2, oc)[0]
        t_or = time.perf_counter() - t
        assert rc == 0 and oc.point_count() == cnt, (oc.point_count(), cnt)
        oc.free()
        best = min(times[1:])
        out[kind] = {"product_first_call_s": times[0], "product_s": best, "product_Mpts_s": n / best / 1e6, "oracle_1thread_s": t_or,
                     "oracle_Mpts_s": n / t_or / 1e6, "count": cnt}
    print(json.dumps(out, indent=1))
