"""bench.py and __graft_entry__.smoke() on a GPU: the printed JSON line follows the driver's contract
(small sizes here; the real run uses the defaults)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check(line, n_gpus):
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "Mpoints/s" and d["n_gpus"] == n_gpus and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "i32"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    assert abs(d["value"] - d["config"]["points"] * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"]) / 1e6) / d["value"] < 1e-9
    assert d["matches"] == d["config"]["points"]  # XL contains every generated point
    assert "traffic_source" in r and r["kernel"] == "k_bounds_count_batch_pipe<2>" and len(r["kernel_source_id"]) == 16
    assert (r["traffic"] is None) == (r["traffic_source"] is None or "not reported" in r["traffic_source"])
    assert len(d["scanned_per_rank"]) == n_gpus == len(d["rank_devices"]) and sum(d["scanned_per_rank"]) == d["config"]["points"]
    return d


def test_bench_json_contract_single_process():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--files", "4",
                        "--points-per-file", "300007", "--cpu-sample-points", "100003"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1  # exactly one JSON line on stdout
    d = _check(lines[0], 1)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mpoints/s" and c["cores"] >= 1 and c["parity_on_sample"] is True
    pb = d["parity_batched_on_sample"]  # the TIMED kernel on a selective box (ca13_L) against the oracle
    assert pb["equal"] is True and pb["gpu"] == pb["oracle"] and pb["kernel"] == d["roofline"]["kernel"] and pb["query"] == "ca13_L"
    assert d["rccl_ranks"] == 0
    # the other configs and the PCIe-inclusive path, measured after the timed region ("as a separate line")
    sec = d["secondary"]
    for key in ("end_to_end", "config2", "config3", "config4"):
        assert key in sec, key
        assert sec[key]["kernels"] and sec[key]["what"]
    e = sec["end_to_end"]
    assert e["bound"] == "pcie" and e["bytes"] == 12 * 4 * 100003 and e["GBps"] > 0 and e["matches"] == 4 * 100003
    assert abs(e["frac_of_pcie_spec"] - e["GBps"] / e["pcie_spec_GBps"]) < 1e-12
    c3 = sec["config3"]
    assert c3["class_histogram_sums_to_n"] is True and c3["class_19"] == 0 and c3["algorithmic_bytes"] == c3["points"] and 0 < c3["matches"] < c3["points"]
    c4 = sec["config4"]
    for cell in ("density_10", "density_100"):
        g = c4[cell]
        assert 0 < g["cells"] <= 300007 and g["algorithmic_bytes"] == 12 * 300007 and g["event_ms"] > 0 and g["wall_ms"] > 0
        assert abs(g["frac"] - g["GBps"] / 8000.0) < 1e-12
    assert c4["density_100"]["cells"] <= c4["density_10"]["cells"]
    c2 = sec["config2"]
    assert c2["halves_sum_to_whole"] is True and c2["algorithmic_bytes"] == 12 * c2["points"]
    assert sec["seconds_spent"] < 60


def test_bench_under_torch_distributed_run_exercises_rccl():
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
                        "--warmup", "1", "--files", "4", "--points-per-file", "300007", "--no-cpu-baseline"],
                       capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = _check(lines[0], 1)
    assert d["cpu_baseline"] is None and d["rccl_ranks"] == 1


def test_bench_two_rank_rehearsal_shards_files_and_sums_counts():
    """The N > 1 code path of bench.py (file sharding, one all-reduce per query enqueued asynchronously, max-over-ranks
    timing) with two ranks sharing the one GPU of the test box; gloo carries the all-reduce there (RCCL refuses two ranks
    on one device), everything else is the path the driver launches on 2/4/8 GPUs."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29543", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--files", "5", "--points-per-file", "300007", "--no-cpu-baseline",
                        "--rehearse-on-one-gpu"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1  # rank 0 only
    d = _check(lines[0], 2)
    assert d["config"]["points"] == 5 * 300007 and d["scaling"] == "strong"
    assert sorted(d["scanned_per_rank"]) == [2 * 300007, 3 * 300007]  # file i -> rank i % 2


def test_smoke_entry_point():
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "smoke ok" in r.stdout
