"""Manual: throughput of the device LZ4 inflater (one wave per frame) by kind of content, next to the host
reader on one thread.  Uses the oracle's test-side compressor, hence under tests/."""
import ctypes as C
import importlib
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
import _oracle
import test_gpu_lz4 as T

pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
binding = importlib.import_module("adhoc-queries-pointclouds_amd.binding")
o = _oracle.Oracle()
T.XXH = o.xxh32
host = C.CDLL(os.path.join(T.PKG, "libpcq_query.so"))
host.pcq_query_lz4_frame_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]
rng = np.random.default_rng(1)
N = 4_000_000
kinds = {
    "random bytes (stored blocks)": rng.integers(0, 256, N, dtype=np.uint8).tobytes(),
    "positions, uniform in a 50 m box at 1 mm (short sequences)": rng.integers(-25000, 25000, size=(N // 12, 3)).astype("<i4").tobytes(),
    "positions, scan-line coherent": np.cumsum(rng.integers(-40, 40, size=(N // 12, 3)), axis=0).astype("<i4").tobytes(),
    "classification bytes": rng.choice(np.array([1, 2, 2, 2, 5, 6], dtype=np.uint8), N).tobytes(),
    "text (long matches)": (b"the quick brown fox jumps over the lazy dog. " * (N // 45 + 1))[:N],
    "zeros": bytes(N),
}
with pkg.Context(0) as ctx:
    for name, data in kinds.items():
        frame = o.lz4f_compress(data, 4, 4)
        nseq = None
        frames = 64
        cases = [(frame, len(data))] * frames
        t = time.perf_counter()
        res = T.run_jobs(ctx, cases)
        dt_total = time.perf_counter() - t
        assert all(st == 0 and got == data for st, got in res[:2])
        # kernel-only: time a second run of just the launch through the binding
        d_src, d_dst = ctx.alloc(len(frame) * frames + 64), ctx.alloc(len(data) * frames + 64)
        ctx.to_device(d_src, np.frombuffer(frame * frames, dtype=np.uint8))
        d = T.descriptor(frame)
        jobs = (binding.Lz4Job * frames)()
        for i in range(frames):
            jobs[i].src = d_src + i * len(frame) + d["payload"]
            jobs[i].src_len = len(frame) - d["payload"]
            jobs[i].dst = d_dst + i * len(data)
            jobs[i].need = len(data)
            jobs[i].block_size_id = d["bsid"]
            jobs[i].independent_blocks = d["independent"]
            jobs[i].has_content_size = d["has_size"]
            jobs[i].content_size = d["content_size"]
        best = 1e9
        for _ in range(3):
            t = time.perf_counter()
            ctx.lz4_inflate(jobs)
            best = min(best, time.perf_counter() - t)
        ctx.free(d_src), ctx.free(d_dst)
        out = np.zeros(len(data), dtype=np.uint8)
        t = time.perf_counter()
        rc = host.pcq_query_lz4_frame_decode(frame, len(frame), len(data), 4, out.ctypes.data, len(data))
        th = time.perf_counter() - t
        assert rc == 0
        print(f"{name:60s} ratio {len(frame) / len(data):5.2f} | device: {len(data) / best / 1e6:8.1f} MB/s per frame, "
              f"{frames * len(data) / best / 1e9:6.2f} GB/s for {frames} frames | host reader, 1 thread: {len(data) / th / 1e6:8.1f} MB/s", flush=True)
