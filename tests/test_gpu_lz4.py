"""GPU: the device LZ4 inflater (csrc/lz4_inflate.hip, pcq_lz4_inflate_dev) against the host reader.

The kernel is a fast path: whatever it reports as done (status 0) must be byte-identical to what the
host reader — itself pinned against the real liblz4 in test_lz4_lazer.py — returns without error; anything
it is unsure about it must hand back (status 1).  Intact frames must actually take the fast path.
"""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

import _lz4ref
from test_lz4_lazer import contents

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "adhoc-queries-pointclouds_amd")
binding = importlib.import_module("adhoc-queries-pointclouds_amd.binding")
REAL = _lz4ref.load()


@pytest.fixture(scope="module")
def host_reader():
    lib = C.CDLL(os.path.join(PKG, "libpcq_query.so"))
    lib.pcq_query_lz4_frame_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]

    def decode(frame: bytes, need: int, unit: int):
        out = np.zeros(max(need, 1), dtype=np.uint8)
        rc = lib.pcq_query_lz4_frame_decode(frame, len(frame), need, unit, out.ctypes.data, max(need, 1))
        return (out[:need].tobytes(), 0) if rc == 0 else (None, rc)

    return decode


XXH = None  # set by the fixture below: the frame descriptor's checksum byte is verified on the host


@pytest.fixture(scope="module", autouse=True)
def _xxh(oracle):
    global XXH
    XXH = oracle.xxh32


def descriptor(frame: bytes):
    """What the host layer extracts before handing a frame to the device; None if the descriptor is not plain."""
    if len(frame) < 7 or frame[:4] != bytes.fromhex("04224d18"):
        return None
    flg, bd = frame[4], frame[5]
    if (flg >> 6) != 1 or flg & 0x02 or bd & 0x8F or ((bd >> 4) & 7) < 4:
        return None
    p = 6
    size = 0
    if flg & 0x08:
        if len(frame) < p + 8:
            return None
        size = int.from_bytes(frame[p:p + 8], "little")
        p += 8
    if flg & 0x01:
        p += 4
    if len(frame) < p + 1 or frame[p] != (XXH(frame[4:p]) >> 8) & 0xFF:
        return None
    return dict(payload=p + 1, bsid=(bd >> 4) & 7, independent=(flg >> 5) & 1, block_checksum=(flg >> 4) & 1,
                has_size=(flg >> 3) & 1, content_size=size)


def run_jobs(ctx, cases):
    """cases: list of (frame, need).  Returns list of (status, bytes or None)."""
    total_src = sum(len(f) for f, _ in cases) + 64
    total_dst = sum(n for _, n in cases) + 64
    d_src, d_dst = ctx.alloc(total_src), ctx.alloc(total_dst)
    ctx.memset(d_dst, 0xEE, total_dst)
    blob = np.frombuffer(b"".join(f for f, _ in cases) + bytes(64), dtype=np.uint8)
    ctx.to_device(d_src, blob)
    jobs = (binding.Lz4Job * len(cases))()
    so = do = 0
    meta = []
    for i, (frame, need) in enumerate(cases):
        d = descriptor(frame)
        meta.append((d, do))
        if d is not None:
            jobs[i].src = d_src + so + d["payload"]
            jobs[i].src_len = len(frame) - d["payload"]
            jobs[i].dst = d_dst + do
            jobs[i].need = need
            jobs[i].content_size = d["content_size"]
            jobs[i].block_size_id = d["bsid"]
            jobs[i].independent_blocks = d["independent"]
            jobs[i].block_checksum = d["block_checksum"]
            jobs[i].has_content_size = d["has_size"]
        so += len(frame)
        do += need
    ctx.lz4_inflate(jobs)
    out = np.zeros(total_dst, dtype=np.uint8)
    ctx.to_host(out, d_dst)
    ctx.free(d_src), ctx.free(d_dst)
    res = []
    for i, (frame, need) in enumerate(cases):
        d, at = meta[i]
        st = jobs[i].status if d is not None else 1
        res.append((st, out[at:at + need].tobytes() if st == 0 else None))
    assert np.all(out[do:] == 0xEE)  # nothing written behind the last destination
    return res


def test_intact_frames_take_the_device_path(gpu_ctx, oracle, host_reader):
    cases, expect = [], []
    for name, data in contents().items():
        for flags in (0, 1, 4, 8, 1 | 4 | 8, 16, 16 | 4):
            for bid in (4, 5, 7):
                frame = oracle.lz4f_compress(data, flags, bid)
                for need in (len(data), max(1, len(data) // 3), 1):
                    cases.append((frame, need))
                    expect.append(data[:need])
    if REAL is not None:
        for name, data in contents().items():
            for kw in (dict(), dict(independent=True, content_size=True), dict(block_id=6, content_checksum=False), dict(level=9)):
                frame = REAL.compress_frame(data, **kw)
                cases.append((frame, len(data)))
                expect.append(data)
    res = run_jobs(gpu_ctx, cases)
    for (st, got), want, (frame, need) in zip(res, expect, cases):
        assert st == 0, (len(frame), need)
        assert got == want


def test_block_checksum_frames_are_left_to_the_host(gpu_ctx, oracle):
    data = contents()["text"]
    res = run_jobs(gpu_ctx, [(oracle.lz4f_compress(data, 2, 4), len(data)), (oracle.lz4f_compress(data, 2 | 4, 4), 100)])
    assert [st for st, _ in res] == [1, 1]


def test_damaged_frames_either_match_the_host_reader_or_are_handed_back(gpu_ctx, oracle, host_reader):
    rng = np.random.default_rng(0xD3)
    c = contents()
    base = []
    for data in (c["text"][:140_000], c["xyz"][:100_000], c["random"][:66_000], b"abcabcabc" * 30, bytes(70_000), c["period3"][:131_077]):
        for flags in (0, 1, 4, 8, 4 | 8, 16):
            base.append((data, oracle.lz4f_compress(data, flags, 4)))
    cases = []
    for it in range(1500):
        data, frame = base[int(rng.integers(0, len(base)))]
        fr = bytearray(frame)
        k = int(rng.integers(0, 6))
        if k == 0:
            for _ in range(int(rng.integers(1, 4))):
                fr[int(rng.integers(0, len(fr)))] ^= 1 << int(rng.integers(0, 8))
        elif k == 1:
            fr = fr[:int(rng.integers(0, len(fr) + 1))]
        elif k == 2:
            fr[int(rng.integers(4, min(24, len(fr))))] = int(rng.integers(0, 256))
        elif k == 3:
            a = int(rng.integers(7, len(fr) - 1))
            fr = fr[:a] + fr[min(len(fr), a + int(rng.integers(1, 64))):]
        elif k == 4:
            a = int(rng.integers(7, len(fr) - 1))
            junk = bytes([255] * int(rng.integers(1, 40))) if it % 2 else rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
            fr = fr[:a] + junk + fr[a:]
        need = int([len(data), len(data) // 2 + 1, 65536, 65537, 1, len(data) + 5][int(rng.integers(0, 6))])
        cases.append((bytes(fr), need))
    res = run_jobs(gpu_ctx, cases)
    handled = 0
    for (st, got), (frame, need) in zip(res, cases):
        if st != 0:
            continue
        handled += 1
        for unit in (4, 2, 1):  # whatever the device calls done, the reference's reader would have produced, in any read size
            want, rc = host_reader(frame, need, unit)
            assert rc == 0 and got == want, (len(frame), need, unit, rc)
    assert 100 < handled < len(cases) - 300  # both outcomes occur


def test_many_jobs_and_large_blobs(gpu_ctx, oracle):
    """A LAZER-shaped batch: hundreds of frames of a megabyte each, inflated in one launch."""
    rng = np.random.default_rng(5)
    xyz = np.cumsum(rng.integers(-300, 300, size=(90_000, 3)), axis=0).astype("<i4").tobytes()  # ~1 MB, mildly compressible
    cls = rng.choice(np.array([1, 2, 2, 2, 5, 6], dtype=np.uint8), 90_000).tobytes()
    frames = [oracle.lz4f_compress(xyz, 4, 4), oracle.lz4f_compress(cls, 4, 4), oracle.lz4f_compress(bytes(540_000), 4, 4)]
    cases = [(frames[i % 3], [len(xyz), len(cls), 540_000][i % 3]) for i in range(300)]
    res = run_jobs(gpu_ctx, cases)
    for i, (st, got) in enumerate(res):
        assert st == 0
        assert got == [xyz, cls, bytes(540_000)][i % 3]
