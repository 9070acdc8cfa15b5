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


PCIE_SPEC_GBS = 63.0      # PCIe Gen5 x16 host link, /opt/skills/guides/MI355X_MICROARCH.md (spec)
PINNED_COPY_GBS = 57.5    # what a pinned hipMemcpyAsync reaches on this host link (profiles/r03_hip_startup.log)


def secondary_measurements(ctx, pkg, binding, specs_mod, torch, dev, tstream, stream, blocks, mine, all_specs, images, sample, bmin, bmax,
                           full_size):
    """The other BASELINE configs and the PCIe-inclusive file path, after the timed region and outside `ms_per_step`
    ("as a separate line").  Every figure names its kernels and its algorithmic bytes; counts are checked."""
    out = {}

    def ev_pair():
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    # (i) end to end from host memory: the CPU-baseline sample (16 x 20 M points, LAST positions blocks in pageable host
    # memory, as a file's mapping is) through pcq_scan_host — pinned double-buffered staging, hipMemcpyAsync against the
    # kernels of the previous chunk.  PCIe-bound: reported as GB/s of the host link, never as an HBM fraction.
    if images is not None:
        def host_pass():
            total, nbytes = 0, 0
            for s_, im in zip(sample, images):
                h = specs_mod.header_fields(s_)
                if not specs_mod.aabb_intersects(h["min"], h["max"], bmin, bmax):
                    continue
                lmin, lmax = pkg.box_to_local(bmin, bmax, h["scale"], h["offset"])
                cols = binding.make_columns(xyz=im.ctypes.data + 227, n=h["n"], scale=h["scale"], offset=h["offset"])
                cc = ctx.count_collector()
                ctx.scan_host(cols, pkg.Predicate.bounds(lmin, lmax), cc)
                total += cc.point_count()
                cc.free()
                nbytes += 12 * h["n"]
            return total, nbytes
        host_pass()  # the pinned ring and the copy helpers exist after the first pass
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            cnt, nbytes = host_pass()
            times.append(time.perf_counter() - t0)
        t = sorted(times)[1]
        out["end_to_end"] = {
            "what": "pcq_scan_host over the CPU-baseline sample (pageable host memory -> pinned staging -> HBM -> K1), bounds "
                    "query of the headline, count read back per file; median of 3 passes",
            "kernels": ["k_bounds_count_w1_pipe<2>"], "bytes": nbytes, "algorithmic_bytes_per_point": 12,
            "seconds": t, "GBps": nbytes / t / 1e9, "Mpoints_per_s": nbytes / 12 / t / 1e6, "matches": cnt,
            "bound": "pcie", "pcie_spec_GBps": PCIE_SPEC_GBS, "frac_of_pcie_spec": nbytes / t / 1e9 / PCIE_SPEC_GBS,
            "pinned_copy_ceiling_GBps": PINNED_COPY_GBS, "frac_of_pinned_copy_ceiling": nbytes / t / 1e9 / PINNED_COPY_GBS,
        }

    # (ii) BASELINE config 3: synthetic doc (8 x 106.75 M points) --class 6, the class blocks resident in HBM, one batched
    # launch per query (K2).  Checked against the class histogram: the counts of every class the generator draws sum to N,
    # class 19 does not occur (run_query_experiments.rs:332-343).
    doc = specs_mod.synth_doc() if full_size else specs_mod.synth_doc(points_per_file=2_000_003)
    cls_blocks, dcols = [], []
    for s_ in doc:
        t = torch.empty(int(s_.n), dtype=torch.uint8, device=dev)
        ctx.synth_fill(s_, 0, int(s_.n), None, t.data_ptr(), stream)
        cls_blocks.append(t)
        h = specs_mod.header_fields(s_)
        dcols.append(binding.make_columns(cls=t.data_ptr(), n=h["n"], scale=h["scale"], offset=h["offset"]))
    ndoc = sum(int(s_.n) for s_ in doc)
    slots = torch.zeros((16, 2), dtype=torch.int64, device=dev)
    hist = {}
    for k, c in enumerate((1, 2, 5, 6, 7, 9, 19, 0)):
        ctx.scan_dev_count_batch(dcols, [pkg.Predicate.classification(c)] * len(dcols), slots[k].data_ptr(), stream)
    torch.cuda.synchronize()
    for k, c in enumerate((1, 2, 5, 6, 7, 9, 19, 0)):
        hist[c] = int(slots[k, 0].item())
    if sum(hist.values()) != ndoc or hist[19] != 0 or hist[0] != 0:
        raise SystemExit(f"PARITY FAILURE config 3: class histogram {hist} does not sum to {ndoc}")
    evs = []
    for k in range(8):
        e0, e1 = ev_pair()
        e0.record(tstream)
        ctx.scan_dev_count_batch(dcols, [pkg.Predicate.classification(6)] * len(dcols), slots[8 + k].data_ptr(), stream)
        e1.record(tstream)
        evs.append((e0, e1))
    torch.cuda.synchronize()
    if any(int(slots[8 + k, 0].item()) != hist[6] for k in range(8)):
        raise SystemExit("PARITY FAILURE config 3: repeated class-6 counts differ")
    ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs[2:])
    ms = ms[len(ms) // 2]
    out["config3"] = {
        "what": f"synth-doc: {len(doc)} LAST files x {int(doc[0].n)} points, --class 6, class blocks resident, one batched launch",
        "kernels": ["k_class_count_batch_pipe<4>", "k_finish_count"], "points": ndoc, "algorithmic_bytes": ndoc,
        "matches": hist[6], "class_histogram_sums_to_n": True, "class_19": hist[19],
        "ms": ms, "GBps": ndoc / (ms * 1e-3) / 1e9, "frac": ndoc / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "Mpoints_per_s": ndoc / (ms * 1e-3) / 1e6,
    }
    del cls_blocks, dcols

    # (iii) BASELINE config 4: ca13 --bounds XL + --density 10 (and 100) on ONE 163 M-point file (per-file grids, main.rs:156):
    # scan (pass 0) + fold until the number of cells is known, wall clock between two device synchronisations and HIP events
    # on the context's stream.  frac = 12 B x N / t / 8 TB/s (the partition and fold traffic is overhead).
    fidx = 5 if 5 in mine else mine[0]
    spec = all_specs[fidx]
    n = int(spec.n)
    xyz = blocks[mine.index(fidx)]
    cls = torch.empty(n, dtype=torch.uint8, device=dev)
    ctx.synth_fill(spec, 0, n, None, cls.data_ptr(), stream)
    torch.cuda.synchronize()
    h = specs_mod.header_fields(spec)
    lmin, lmax = pkg.box_to_local(bmin, bmax, h["scale"], h["offset"])
    gcols = binding.make_columns(xyz=xyz.data_ptr(), cls=cls.data_ptr(), n=n, scale=h["scale"], offset=h["offset"])
    known = {10.0: 122_507_711, 100.0: 1_872_525} if (n == 163_000_000 and fidx == 5) else {}
    cstream = torch.cuda.ExternalStream(ctx.stream_handle(), device=dev)
    grid = {}
    for cell in (10.0, 100.0):
        runs = []
        for _ in range(4):
            g = ctx.grid_collector(bmin, bmax, cell)
            ctx.synchronize()
            e0, e1 = ev_pair()
            t0 = time.perf_counter()
            e0.record(cstream)
            ctx.scan_dev(gcols, pkg.Predicate.bounds(lmin, lmax), g)
            cells = g.point_count()  # folds
            e1.record(cstream)
            ctx.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            runs.append((wall, e0.elapsed_time(e1), cells))
            g.free()
        cells = runs[-1][2]
        if any(r[2] != cells for r in runs) or (cell in known and cells != known[cell]):
            raise SystemExit(f"PARITY FAILURE config 4: cells at {cell} m: {[r[2] for r in runs]}, known answer {known.get(cell)}")
        wall = sorted(r[0] for r in runs[1:])[1]
        evms = sorted(r[1] for r in runs[1:])[1]
        grid[f"density_{int(cell)}"] = {
            "cells": cells, "cells_known_answer": known.get(cell), "wall_ms": wall, "event_ms": evms,
            "algorithmic_bytes": 12 * n, "GBps": 12 * n / (evms * 1e-3) / 1e9, "frac": 12 * n / (evms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "Mpoints_per_s": n / (evms * 1e-3) / 1e6,
        }
    out["config4"] = {
        "what": f"synth-ca13 file {fidx} ({n} points, generator order), --bounds ca13_XL --density 10 / 100: scan + fold of one "
                "per-file grid, median of 3 after a warm-up",
        "kernels": ["k_p0_part", "k_dir_transpose", "k_bin_prefix", "k_level2", "k_fold_dense", "k_fold"], **grid,
    }
    del cls

    # (iv) BASELINE config 2: synthetic navvis (1 file x 56.2 M points) --bounds S, count, positions resident (K1, per-file kernel).
    nav = specs_mod.synth_navvis()[0] if full_size else specs_mod.synth_navvis(points_per_file=2_000_003)[0]
    nn = int(nav.n)
    t = torch.empty(12 * nn, dtype=torch.uint8, device=dev)
    ctx.synth_fill(nav, 0, nn, t.data_ptr(), None, stream)
    hn = specs_mod.header_fields(nav)
    nb0, nb1 = specs_mod.box("navvis_S")
    lmin, lmax = pkg.box_to_local(nb0, nb1, hn["scale"], hn["offset"])
    ncols = binding.make_columns(xyz=t.data_ptr(), n=nn, scale=hn["scale"], offset=hn["offset"])
    slot = torch.zeros((12, 2), dtype=torch.int64, device=dev)
    evs = []
    for k in range(12):
        cc = ctx.count_collector(device_counter=slot[k].data_ptr())
        e0, e1 = ev_pair()
        e0.record(tstream)
        ctx.scan_dev(ncols, pkg.Predicate.bounds(lmin, lmax), cc, stream)
        e1.record(tstream)
        evs.append((e0, e1))
        cc.free()
    torch.cuda.synchronize()
    got = [int(slot[k, 0].item()) for k in range(12)]
    # the box is axis-aligned in integer space: its count is the sum of the counts of its two halves along x (exact partition)
    mid = (max(lmin[0], -2 ** 31) + min(lmax[0], 2 ** 31 - 1)) // 2
    halves = torch.zeros((2, 2), dtype=torch.int64, device=dev)
    for k, (a, b) in enumerate((([lmin[0], lmin[1], lmin[2]], [mid, lmax[1], lmax[2]]), ([mid + 1, lmin[1], lmin[2]], [lmax[0], lmax[1], lmax[2]]))):
        cc = ctx.count_collector(device_counter=halves[k].data_ptr())
        ctx.scan_dev(ncols, pkg.Predicate.bounds(a, b), cc, stream)
        cc.free()
    torch.cuda.synchronize()
    if len(set(got)) != 1 or int(halves[0, 0].item()) + int(halves[1, 0].item()) != got[0]:
        raise SystemExit(f"PARITY FAILURE config 2: navvis S counts {sorted(set(got))}, halves {halves[:, 0].tolist()}")
    ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs[2:])
    ms = ms[len(ms) // 2]
    out["config2"] = {
        "what": f"synth-navvis: 1 LAST file x {nn} points, --bounds navvis_S, count, positions resident, per-file kernel",
        "kernels": ["k_bounds_count_w1_pipe<2>", "k_finish_count"], "points": nn, "algorithmic_bytes": 12 * nn, "matches": got[0],
        "halves_sum_to_whole": True, "ms": ms, "GBps": 12 * nn / (ms * 1e-3) / 1e9, "frac": 12 * nn / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "Mpoints_per_s": nn / (ms * 1e-3) / 1e6,
        "note": "a 0.67 GB file is a 0.1 ms launch: launch-bound, not stream-bound",
    }
    return out


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
    ap.add_argument("--no-secondary", action="store_true", help="skip the untimed measurements of configs 2-4 and of the host path")
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

    # Everything else the driver should see, measured AFTER the timed region, outside `ms_per_step`:
    # the PCIe-inclusive file path and BASELINE configs 2, 3, 4.  Rank 0 at N = 1 only.
    if rank == 0 and world == 1 and not args.no_secondary:
        t_sec = time.perf_counter()
        have_sample = not args.no_cpu_baseline
        result["secondary"] = secondary_measurements(
            ctx, pkg, binding, specs_mod, torch, dev, tstream, stream, blocks, mine, all_specs,
            images if have_sample else None, sample if have_sample else None, bmin, bmax,
            full_size=args.points_per_file >= 100_000_000)
        result["secondary"]["seconds_spent"] = time.perf_counter() - t_sec

    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
