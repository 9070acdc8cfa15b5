#!/usr/bin/env python3
"""bench.py — the north-star measurement: ca13 XL bounds query (count), 1/2/4/8 MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (config.workload): synthetic ca13 — 16 LAST files x 163,000,000 points (2,608 Mpoints,
31.3 GB of positions), SURVEY.md §8(d) — query `--bounds` ca13-XL (run_query_experiments.rs:141-144),
count only.  The positions blocks are generated directly in HBM by pcq_synth_fill_dev (bit-identical
to the host generator) and are resident before the timed region starts.

A step = one query over the whole dataset: per file the header early-out (last.rs:92-94) and the
f64 -> local integer box (last.rs:98-109) on the host, one batched bounds_count launch over this
rank's files, the global sum of the match counts (main.rs:164-180; an RCCL all-reduce when N > 1) and
the 8-byte result read back to the host.  Files are the independent units (main.rs:153-161): with N
ranks, file i belongs to rank i % N; the dataset is fixed, so this is strong scaling.

One JSON line is printed by rank 0 (see the contract in the task statement).  `roofline` is computed
from HIP events recorded around every launch of the dominant kernel inside the timed region, on the
stream the kernel runs on; `cpu_baseline` times the oracle (the C restatement of the reference's
`--optimized --parallel` loop; "port": the reference itself is Rust and cannot be built here) on a
bounded sample of the same synthetic files on the host cores.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL across processes needs on this driver

import torch  # noqa: E402  (device memory, streams, torch.distributed — plumbing only)
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def kernel_source_id():
    """Identity of the timed kernel's source (stable across rebuilds): sha256 of the files it is compiled from."""
    import hashlib
    h = hashlib.sha256()
    for rel in ("adhoc-queries-pointclouds_amd/csrc/scan_count.hip", "adhoc-queries-pointclouds_amd/csrc/pcq_internal.h"):
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()[:16]
BYTES_PER_POINT = 12   # bounds count reads N x {i32 x,y,z}


def cpu_baseline(specs_mod, bmin, bmax, sample_points_per_file, nfiles):
    """Oracle on the host cores: one thread per file (rayon par_iter, main.rs:153-161)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle  # the checker / CPU baseline, never the product path
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"], check=True, capture_output=True)
    oracle = _oracle.Oracle()
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    threads = min(nfiles, cores)
    sample = specs_mod.synth_ca13(points_per_file=sample_points_per_file, files=nfiles)
    images = [oracle.synth_image(s, transposed=True, threads=cores) for s in sample]
    total_pts = sum(int(s.n) for s in sample)
    times, count = [], None
    for _ in range(5):  # run_query_experiments.rs:412 — 5 runs, median
        t0 = time.perf_counter()
        count = oracle.count_files_parallel(images, 0, bmin, bmax, 0, threads)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return oracle, threads, {"value": total_pts / med / 1e6, "unit": "Mpoints/s", "cores": threads, "kind": "port",
            "sample": f"oracle (C restatement of last.rs:46-166 + CountCollector), {nfiles} synthetic ca13 LAST files x "
                      f"{sample_points_per_file} points in host memory, warm, one thread per file, median of 5; "
                      f"matches {count}/{total_pts}; host has {cores} cores"}, images, sample, count


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points-per-file", type=int, default=163_000_000)
    ap.add_argument("--files", type=int, default=16)
    ap.add_argument("--query", type=str, default="ca13_XL")
    ap.add_argument("--cpu-sample-points", type=int, default=20_000_000, help="points per file of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-file-launch", action="store_true", help="one launch per file instead of one batched launch")
    ap.add_argument("--blocks-per-cu", type=int, default=0, help="tuning: persistent blocks per CU (0 = library default)")
    ap.add_argument("--sync-each-step", action="store_true",
                    help="read every query's answer back before the next query starts (query latency); default: the K "
                         "queries are enqueued back to back, each into its own result slot, and all K answers are read "
                         "and checked after the timed region (query throughput)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a single-GPU box: every rank uses device 0 and the gloo backend "
                         "(RCCL refuses two ranks on one device); numbers from such a run are not a measurement")
    args = ap.parse_args()

    # stdout carries exactly ONE JSON line: native libraries (the RCCL banner, rocprof) write to fd 1,
    # so everything else is routed to stderr and the result is written to the saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
        args.gpus = world

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ  # under torch.distributed.run the RCCL path is exercised even at N=1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    rank_devices = [local_rank]
    if use_dist:
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"torch.distributed world size {dist.get_world_size()} != --gpus {args.gpus}")
        # one rank per GPU: two ranks on one device would halve the per-GPU rate and report it as scaling
        ident = (os.uname().nodename, local_rank, str(getattr(torch.cuda.get_device_properties(dev), "uuid", "")))
        gathered = [None] * world
        dist.all_gather_object(gathered, ident)
        if not args.rehearse_on_one_gpu and len(set(gathered)) != world:
            raise SystemExit(f"ranks share a device: {gathered}")
        rank_devices = [g[1] for g in gathered]

    pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
    binding = importlib.import_module("adhoc-queries-pointclouds_amd.binding")
    specs_mod = importlib.import_module("adhoc-queries-pointclouds_amd.synth_specs")
    sharding = importlib.import_module("adhoc-queries-pointclouds_amd.sharding")

    all_specs = specs_mod.synth_ca13(points_per_file=args.points_per_file, files=args.files)
    mine = sharding.assign_files(len(all_specs), world, rank, points=[int(s.n) for s in all_specs])  # LPT; equal files: i -> rank i % N
    bmin, bmax = specs_mod.box(args.query)

    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    ctx = pkg.Context(local_rank)
    info = ctx.device_info()
    if args.blocks_per_cu:
        ctx.set_option("blocks_per_cu", args.blocks_per_cu)

    # ---- dataset: positions blocks resident in HBM (torch owns the memory) -----------------------
    blocks, headers = [], []
    for i in mine:
        s = all_specs[i]
        t = torch.empty(int(s.n) * BYTES_PER_POINT, dtype=torch.uint8, device=dev)
        ctx.synth_fill(s, 0, int(s.n), t.data_ptr(), None, stream)
        blocks.append(t)
        headers.append(specs_mod.header_fields(s))
    torch.cuda.synchronize()

    # one result slot per query: [k, 0] = match count of query k (all-reduced over the ranks)
    answers = torch.zeros((args.warmup + args.steps + 1, 2), dtype=torch.int64, device=dev)
    local_points = sum(h["n"] for h in headers)
    global_points = sum(int(s.n) for s in all_specs)

    pending = []  # all-reduces enqueued by streamed queries

    def query_step(slot, record=None, sync=True):
        """One `--bounds XL --optimized --parallel` count query over the dataset."""
        total = answers[slot]
        cols, preds, scanned = [], [], 0
        for t, h in zip(blocks, headers):
            if not specs_mod.aabb_intersects(h["min"], h["max"], bmin, bmax):  # last.rs:92-94
                continue
            lmin, lmax = pkg.box_to_local(bmin, bmax, h["scale"], h["offset"])  # last.rs:98-109
            cols.append(binding.make_columns(xyz=t.data_ptr(), n=h["n"], scale=h["scale"], offset=h["offset"]))
            preds.append(pkg.Predicate.bounds(lmin, lmax))
            scanned += h["n"]
        if cols:
            if args.per_file_launch:
                cc = ctx.count_collector(device_counter=total.data_ptr())
                for c, p in zip(cols, preds):
                    if record is not None:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(tstream)
                    ctx.scan_dev(c, p, cc, stream)
                    if record is not None:
                        e1.record(tstream)
                        record.append((e0, e1, c.n))
                cc.free()
            else:
                if record is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(tstream)
                ctx.scan_dev_count_batch(cols, preds, total.data_ptr(), stream)
                if record is not None:
                    e1.record(tstream)
                    record.append((e0, e1, scanned))
        # main.rs:164-180: one RCCL all-reduce of the count.  In the streamed mode it is only enqueued: it runs on the
        # communicator's stream behind this query's kernels while the next query's scan already occupies the GPU
        work = sharding.global_count(total, world if not use_dist else max(world, 2), async_op=not sync)
        if not sync and work is not None:
            pending.append(work)
        return (int(total[0].item()) if sync else None), scanned

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    expected, _ = query_step(args.warmup + args.steps)  # the answer, read back once (untimed)
    for k in range(args.warmup):
        query_step(k, sync=args.sync_each_step)
    for w in pending:
        w.wait()
    pending.clear()
    barrier()
    events = []
    t0 = time.perf_counter()
    for k in range(args.steps):
        _, scanned_local = query_step(args.warmup + k, events, sync=args.sync_each_step)
    for w in pending:  # every all-reduce of the timed queries has completed before the clock stops
        w.wait()
    pending.clear()
    if use_dist:
        # Every rank says what it did BEFORE the closing barrier: a scaling run that hangs or fails there can be read from
        # the tail of its stderr (which rank never arrived, on which device, after how much work).
        torch.cuda.synchronize()
        sys.stderr.write("[bench] rank %d/%d local_rank %d device %s uuid %s: scanned %d points per query, %d queries in %.3f s (own clock)\n" % (
            rank, world, local_rank, torch.cuda.get_device_name(dev), str(getattr(torch.cuda.get_device_properties(dev), "uuid", "?")),
            scanned_local, args.steps, time.perf_counter() - t0))
        sys.stderr.flush()
    barrier()
    elapsed = time.perf_counter() - t0
    # every timed query produced the answer (nothing was skipped or cached: each query wrote its own slot)
    got = answers[args.warmup:args.warmup + args.steps, 0].cpu().tolist()
    if any(g != expected for g in got):
        raise SystemExit(f"query answers differ inside the timed region: expected {expected}, got {sorted(set(got))}")
    matches = expected
    if use_dist:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
        per_rank = torch.zeros(world, dtype=torch.int64, device=dev)
        per_rank[rank] = scanned_local
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)  # every rank's share, through the same communicator as the counts
        scanned_per_rank = [int(v) for v in per_rank.cpu().tolist()]
        scanned_global = sum(scanned_per_rank)
    else:
        scanned_global = scanned_local
        scanned_per_rank = [scanned_local]

    # kernel-only roofline (this rank's launches; HIP events on the launch stream)
    launch_ms = [e0.elapsed_time(e1) for e0, e1, _ in events]
    launch_pts = [n for _, _, n in events]
    avg_ms = sum(launch_ms) / max(1, len(launch_ms))
    alg_bytes = BYTES_PER_POINT * (sum(launch_pts) / max(1, len(launch_pts)))
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0

    kernel_name = "k_bounds_count_w1_pipe<2>" if args.per_file_launch else "k_bounds_count_batch_pipe<2>"
    result = None
    if rank == 0:
        # HBM traffic of the timed kernel from PMC counters.  A process cannot profile itself: the figure comes from the
        # committed summary of tools/pmc_traffic.py (bench.py under rocprofv3 --pmc, separate passes) and is reported
        # only if that summary was taken on THIS kernel (same name, same source files, same points per launch).
        traffic, traffic_source = None, None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                same = (pmc.get("points_per_launch") == (sum(launch_pts) // max(1, len(launch_pts)))
                        and kernel_name in str(pmc.get("kernel_name", "")) and pmc.get("kernel_source_id") == kernel_source_id())
                if same:
                    traffic = pmc.get("hbm_bytes_per_launch")
                    traffic_source = f"profiles/pmc_latest.json@{pmc.get('git_head', '?')} (tools/pmc_traffic.py, not measured by this run)"
                elif kernel_name in str(pmc.get("kernel_name", "")) and pmc.get("kernel_source_id") == kernel_source_id():
                    traffic_source = (f"profiles/pmc_latest.json@{pmc.get('git_head', '?')} counted this kernel with {pmc.get('points_per_launch')} points per launch "
                                      f"(traffic / algorithmic {pmc.get('traffic_over_algorithmic')}); this run's launches differ: not reported")
                else:
                    traffic_source = "profiles/pmc_latest.json is from another kernel build: not reported"
            except Exception:
                traffic = None
        result = {
            "metric": "Mpoints/s filtered + achieved HBM GB/s, ca13 XL bounds query, 1/2/4/8 GPUs",
            "value": scanned_global * args.steps / elapsed / 1e6,
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "i32",
            "data": "synthetic",
            "config": {
                "workload": f"synth-ca13: {args.files} LAST files x {args.points_per_file} points "
                            f"({global_points / 1e6:.0f} Mpoints), query --bounds {args.query} count-only, positions resident in HBM",
                "files": args.files,
                "points": global_points,
                "query": args.query,
                "sharding": "file i -> rank i % N, one RCCL all-reduce(sum, u64) per query" if world > 1 else "single GPU",
                "launch": "per-file" if args.per_file_launch else "one batched launch per query",
                "queries": "answer read back before the next query" if args.sync_each_step else
                           "enqueued back to back, one result slot each; all answers read and checked after the timed region",
                "device": info["name"] + " " + info["gcn_arch"],
            },
            "rccl_ranks": dist.get_world_size() if use_dist and not args.rehearse_on_one_gpu else 0,
            "rank_devices": rank_devices,
            "scanned_per_rank": scanned_per_rank,
            "matches": matches,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": kernel_name,
                "kernel_source_id": kernel_source_id(),
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_launch_ms": avg_ms,
                "launches_timed": len(launch_ms),
            },
        }

    # CPU baseline: rank 0 at N=1 only, bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        oracle, threads, base, images, sample, cpu_count = cpu_baseline(specs_mod, bmin, bmax, args.cpu_sample_points, args.files)
        result["cpu_baseline"] = base
        # The TIMED kernel with a selective predicate: one untimed `--bounds ca13_L` query over the sample, resident in HBM,
        # through the same batched launch as the timed region, against the oracle's count for that box.
        lb_min, lb_max = specs_mod.box("ca13_L")
        want_l = oracle.count_files_parallel(images, 0, lb_min, lb_max, 0, threads)
        dcols, dpreds, keep = [], [], []
        for s_, im in zip(sample, images):
            h = specs_mod.header_fields(s_)
            if not specs_mod.aabb_intersects(h["min"], h["max"], lb_min, lb_max):  # last.rs:92-94
                continue
            t = torch.from_numpy(im[227:227 + 12 * h["n"]]).to(dev)  # the positions block (offset_to_point_data = 227)
            keep.append(t)
            lmin, lmax = pkg.box_to_local(lb_min, lb_max, h["scale"], h["offset"])
            dcols.append(binding.make_columns(xyz=t.data_ptr(), n=h["n"], scale=h["scale"], offset=h["offset"]))
            dpreds.append(pkg.Predicate.bounds(lmin, lmax))
        slot = torch.zeros(2, dtype=torch.int64, device=dev)
        if dcols:
            ctx.scan_dev_count_batch(dcols, dpreds, slot.data_ptr(), stream)
        got_l = int(slot[0].item())
        del keep
        result["parity_batched_on_sample"] = {"query": "ca13_L", "kernel": kernel_name, "gpu": got_l, "oracle": int(want_l),
                                              "files_scanned": len(dcols), "equal": bool(got_l == want_l)}
        if got_l != want_l:
            raise SystemExit(f"PARITY FAILURE of the batched kernel on the sample (ca13_L): gpu {got_l} != oracle {want_l}")
        # the same sample through the GPU path must give the same count (bit-exact parity on the bench inputs)
        gpu_total = 0
        for s, im in zip(sample, images):
            h = specs_mod.header_fields(s)
            if not specs_mod.aabb_intersects(h["min"], h["max"], bmin, bmax):
                continue
            lmin, lmax = pkg.box_to_local(bmin, bmax, h["scale"], h["offset"])
            cols = binding.make_columns(xyz=im.ctypes.data + 227, n=h["n"], scale=h["scale"], offset=h["offset"])
            cc = ctx.count_collector()
            ctx.scan_host(cols, pkg.Predicate.bounds(lmin, lmax), cc)
            gpu_total += cc.point_count()
            cc.free()
        result["cpu_baseline"]["gpu_count_on_sample"] = gpu_total
        result["cpu_baseline"]["parity_on_sample"] = bool(gpu_total == cpu_count)
        if gpu_total != cpu_count:
            raise SystemExit(f"PARITY FAILURE on the CPU-baseline sample: gpu {gpu_total} != oracle {cpu_count}")
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
