"""LAZER row, CPU side: the LZ4 Frame readers (oracle restatement and the product's
host decoder) pinned against the real liblz4 the reference's `lz4` crate wraps, the committed golden
frames, and the oracle's LAZER searches against numpy restatements of query/src/search/lazer.rs.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import _lz4ref
from _oracle import ERR_EOF, ERR_HEADER, ERR_PANIC, OK, POINT_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "adhoc-queries-pointclouds_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")

REAL = _lz4ref.load()
needs_liblz4 = pytest.mark.skipif(REAL is None, reason="liblz4.so.1 not on this host")


@pytest.fixture(scope="module")
def product():
    """The product's host-side LZ4 Frame reader through the C view (no GPU call)."""
    lib = C.CDLL(os.path.join(PKG, "libpcq_query.so"))
    lib.pcq_query_lz4_frame_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]
    lib.pcq_query_last_error.restype = C.c_char_p

    def decode(frame: bytes, need: int, unit: int = 0):
        out = np.zeros(max(need, 1), dtype=np.uint8)
        rc = lib.pcq_query_lz4_frame_decode(frame, len(frame), need, unit, out.ctypes.data, max(need, 1))
        return (out[:need].tobytes(), 0) if rc == 0 else (None, rc)

    return decode


def contents():
    rng = np.random.default_rng(20260104)
    xyz = np.cumsum(rng.integers(-40, 40, size=(30000, 3)), axis=0).astype("<i4")  # coherent scan-like positions
    return {
        "one": b"x",
        "short": b"hello hello hello hello hello",
        "zeros": bytes(200_000),
        "random": rng.integers(0, 256, 150_000, dtype=np.uint8).tobytes(),
        "text": (b"the quick brown fox jumps over the lazy dog. " * 4000),
        "xyz": xyz.tobytes(),
        "classes": rng.choice(np.array([1, 2, 2, 2, 5, 6], dtype=np.uint8), 90_000).tobytes(),
        "period3": bytes([1, 2, 3]) * 50_000,
    }


# ---- xxHash32 ------------------------------------------------------------------------------------
def test_xxh32_known_answers(oracle):
    # published xxHash32 (seed 0) values
    assert oracle.xxh32(b"") == 0x02CC5D05
    assert oracle.xxh32(b"a") == 0x550D7456
    assert oracle.xxh32(b"abc") == 0x32D153FF
    assert oracle.xxh32(b"Nobody inspects the spammish repetition") == 0xE2293B2F
    import xxhash
    rng = np.random.default_rng(5)
    for n in (1, 3, 4, 15, 16, 17, 31, 32, 33, 1000, 65537):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert oracle.xxh32(d) == xxhash.xxh32(d).intdigest()


# ---- golden frames made by the real liblz4 (committed, so this runs anywhere) ----------------------------
def test_golden_liblz4_frames(oracle, product):
    with open(os.path.join(GOLDEN, "lz4_frames.json")) as f:
        cases = json.load(f)["frames"]
    assert len(cases) >= 12
    for c in cases:
        frame, content = bytes.fromhex(c["frame"]), bytes.fromhex(c["pattern"]) * c["repeat"]
        for need in {len(content), len(content) // 2, 1}:
            if need == 0:
                continue
            got, rc = oracle.lz4f_decode(frame, need)
            assert rc == 0 and got == content[:need], c["name"]
            got, rc = product(frame, need)
            assert rc == 0 and got == content[:need], c["name"]
        # one byte more than the frame holds: read_exact's UnexpectedEof
        assert oracle.lz4f_decode(frame, len(content) + 1)[1] == ERR_EOF
        assert product(frame, len(content) + 1)[1] == ERR_EOF


# ---- live: every flag combination of the real compressor ----------------------------------------------------
@needs_liblz4
def test_decoders_match_real_liblz4(oracle, product):
    assert REAL.version() >= 10900
    for name, data in contents().items():
        for block_id in (0, 4, 5, 7):
            for independent in (False, True):
                for csum, bsum, csize in ((True, False, False), (False, True, True), (True, True, False), (False, False, False)):
                    for level in (0, 9):
                        if level == 9 and name not in ("text", "xyz"):
                            continue
                        frame = REAL.compress_frame(data, block_id, independent, csum, bsum, csize, level)
                        for need in (len(data), max(1, len(data) // 3)):
                            got, rc = oracle.lz4f_decode(frame, need)
                            assert rc == 0 and got == data[:need], (name, block_id, independent, csum, bsum, csize)
                            got, rc = product(frame, need)
                            assert rc == 0 and got == data[:need], (name, block_id, independent, csum, bsum, csize)


@needs_liblz4
def test_oracle_writer_is_readable_by_real_liblz4(oracle, product):
    """The test-side LZ4 writer (used to build LAZER files) emits frames the real library accepts."""
    for name, data in contents().items():
        for flags in (0, 1, 2, 4, 8, 15, 16, 16 | 4, 31):
            for block_id in (4, 6):
                frame = oracle.lz4f_compress(data, flags, block_id)
                assert REAL.read_exact(frame, len(data)) == data, (name, flags, block_id)
                if name in ("zeros", "text", "period3", "xyz") and not flags & 16:
                    assert len(frame) < len(data) * 0.8, (name, len(frame))  # it does compress
                got, rc = product(frame, len(data))
                assert rc == 0 and got == data
                got, rc = oracle.lz4f_decode(frame, len(data))
                assert rc == 0 and got == data


def _classify(fn):
    try:
        fn()
        return OK
    except _lz4ref.UnexpectedEof:
        return ERR_EOF
    except _lz4ref.LZ4Error:
        return ERR_HEADER


@needs_liblz4
def test_damaged_frames_fail_like_the_real_reader(oracle, product):
    """What a streaming read of exactly `need` bytes notices — and what it never gets to see."""
    data = contents()["text"][:150_000]  # three 64 KiB blocks
    frame = REAL.compress_frame(data, 4, False, True, True, True)
    n = len(data)

    def outcome(fr, need, same=True):
        # the reference pulls read_i32 / read_u16 / read_u8 sized pieces (lazer_reader.rs:598-600, :693, :665)
        res = {}
        for unit in (4, 2, 1, 0):
            real = _classify(lambda: REAL.read_exact(fr, need, unit))
            o = oracle.lz4f_decode(fr, need, unit)[1]
            p = product(fr, need, unit)[1]
            assert o == real and p == real, (need, unit, real, o, p)
            res[unit] = real
        if same:
            assert len(set(res.values())) == 1, res
            return res[4]
        return res

    # the intact frame; one byte too many -> UnexpectedEof
    assert outcome(frame, n) == OK
    assert outcome(frame, n + 1) == ERR_EOF
    # corrupt content checksum (last 4 bytes): unseen when exactly n bytes are pulled, an LZ4 error one byte later
    bad = frame[:-1] + bytes([frame[-1] ^ 0x55])
    assert outcome(bad, n) == OK
    assert outcome(bad, n + 1) == ERR_HEADER
    # EndMark and checksum cut off, i.e. the input ends exactly behind the last block: lz4::Decoder stops
    # polling liblz4 once it has no input left, so of that block only the read that inflated it delivers
    last_block_start = 2 * 65536
    res = outcome(frame[:-8], n, same=False)
    assert res == {4: ERR_EOF, 2: ERR_EOF, 1: ERR_EOF, 0: OK}
    assert outcome(frame[:-8], last_block_start) == OK
    assert outcome(frame[:-8], last_block_start + 1, same=False) == {4: OK, 2: OK, 1: OK, 0: OK}
    assert outcome(frame[:-8], last_block_start + 2, same=False) == {4: OK, 2: OK, 1: ERR_EOF, 0: OK}
    assert outcome(frame[:-8], last_block_start + 4, same=False) == {4: OK, 2: ERR_EOF, 1: ERR_EOF, 0: OK}
    assert outcome(frame[:-8], last_block_start + 5, same=False) == {4: ERR_EOF, 2: ERR_EOF, 1: ERR_EOF, 0: OK}
    assert outcome(frame[:-7], n) == OK  # one input byte behind the block keeps the reader going
    assert outcome(frame[:-8], n + 1) == ERR_EOF
    # cut inside the last block: the first two blocks still come out
    assert outcome(frame[:-40], 2 * 65536) == OK
    assert outcome(frame[:-40], 2 * 65536 + 1) == ERR_EOF
    # a flipped byte in the first block breaks its block checksum — nothing comes out
    hit = bytearray(frame)
    hit[40] ^= 0xFF
    assert outcome(bytes(hit), 1) == ERR_HEADER
    # wrong magic, wrong header checksum, wrong version, reserved bit, bad block-size id
    assert outcome(b"\x05" + frame[1:], 1) == ERR_HEADER
    hc = 4 + 2 + 8
    assert outcome(frame[:hc] + bytes([frame[hc] ^ 1]) + frame[hc + 1:], 1) == ERR_HEADER
    for flg_bd in ((0x80 | (frame[4] & 0x3F), frame[5]), (frame[4] | 0x02, frame[5]), (frame[4], 0x30), (frame[4], frame[5] | 0x01)):
        assert outcome(frame[:4] + bytes(flg_bd) + frame[6:], 1) == ERR_HEADER
    # wrong content size in the header (header checksum fixed up): noticed at the EndMark, which the read
    # of the last content byte already reaches; an earlier stop does not
    desc = bytearray(frame[4:hc])
    desc[2:10] = (n + 7).to_bytes(8, "little")
    fixed = frame[:4] + bytes(desc) + bytes([(oracle.xxh32(bytes(desc)) >> 8) & 0xFF]) + frame[hc + 1:]
    assert outcome(fixed, n) == ERR_HEADER
    assert outcome(fixed, n - 1) == OK
    assert outcome(fixed, 65536) == OK
    assert outcome(fixed[:-5], n) == OK  # ... unless the EndMark is not (completely) there to be read
    # the header of the block after the last needed one is looked at: too large a size is an error
    cut = 4 + 2 + 8 + 1
    first = int.from_bytes(frame[cut:cut + 4], "little") & 0x7FFFFFFF
    nxt = cut + 4 + first + 4
    big = frame[:nxt] + (70000).to_bytes(4, "little") + frame[nxt + 4:]
    assert outcome(big, 65536) == ERR_HEADER
    assert outcome(big, 65535) == OK
    # empty input, magic only
    assert outcome(b"", 1) == ERR_EOF
    assert outcome(frame[:4], 1) == ERR_EOF
    assert outcome(frame[:9], 1) == ERR_EOF
    # a skippable frame in front ends the stream for lz4::Decoder (LZ4F_decompress returns 0 after it)
    skip = (0x184D2A50).to_bytes(4, "little") + (3).to_bytes(4, "little") + b"abc"
    assert outcome(skip + frame, 1) == ERR_EOF


def test_zero_byte_request_never_touches_the_frame(oracle, product):
    assert oracle.lz4f_decode(b"garbage", 0) == (b"", 0)
    assert product(b"garbage", 0) == (b"", 0)


# ---- LAZER searches of the oracle against numpy ------------------------------------------------------------
def _spec(pcq, n, fmt, seed=7):
    specs = __import__("importlib").import_module("adhoc-queries-pointclouds_amd.synth_specs")
    s = specs.synth_navvis(points_per_file=n)[0]
    s.seed = seed
    s.format = fmt
    return s


def _columns(oracle, image):
    h = oracle.parse_header(image[:400].tobytes())
    n, otp = h.number_of_points, h.offset_to_point_data
    fmt = h.point_data_record_format
    xyz = image[otp:otp + 12 * n].view("<i4").reshape(-1, 3)
    cls = image[otp + 15 * n: otp + 16 * n]
    if fmt in (2, 3):
        co = {2: 20, 3: 28}[fmt]
        rgb = image[otp + co * n: otp + co * n + 6 * n].view("<u2").reshape(-1, 3)
    else:
        rgb = np.zeros((n, 3), dtype=np.uint16)
    world = np.empty((n, 3))
    for a in range(3):
        world[:, a] = h.offset[a] + h.scale[a] * xyz[:, a].astype(np.float64)
    return h, world, cls, rgb


def _expect(world, cls, rgb, idx):
    out = np.zeros(len(idx), dtype=POINT_DTYPE)
    out["x"], out["y"], out["z"] = world[idx, 0], world[idx, 1], world[idx, 2]
    out["r"], out["g"], out["b"] = rgb[idx, 0], rgb[idx, 1], rgb[idx, 2]
    out["classification"] = cls[idx]
    return out


@pytest.mark.parametrize("fmt", [0, 1, 2, 3])
@pytest.mark.parametrize("block_size", [1, 7, 1000, 5000, 9999])
def test_oracle_lazer_bounds_matches_numpy(oracle, pcq, fmt, block_size):
    n = 5000
    if block_size == 1:
        n = 300
    image = oracle.synth_image(_spec(pcq, n, fmt), transposed=True)
    h, world, cls, rgb = _columns(oracle, image)
    lazer = oracle.lazer_from_last(image, block_size, flags=4 if fmt % 2 else 1 | 2 | 8, block_id=4)
    lo = np.quantile(world, 0.3, axis=0)
    hi = np.quantile(world, 0.8, axis=0)
    inside = np.all((world >= lo) & (world <= hi), axis=1)
    bc = oracle.buffer_collector()
    assert oracle.search_lazer_bounds(lazer, lo, hi, bc) == OK
    got = bc.points()
    bc.free()
    want = _expect(world, cls, rgb, np.flatnonzero(inside))
    assert len(got) == len(want) > 0
    assert got.tobytes() == want.tobytes()
    # boundary points are contained: a box that is exactly one point
    p = world[n // 2]
    cc = oracle.count_collector()
    assert oracle.search_lazer_bounds(lazer, p, p, cc) == OK
    assert cc.point_count() == int(np.all(world == p, axis=1).sum()) >= 1
    cc.free()
    # header-AABB early-out
    cc = oracle.count_collector()
    far = np.array(h.max) + 10.0
    assert oracle.search_lazer_bounds(lazer, far, far + 1.0, cc) == OK
    assert cc.point_count() == 0
    cc.free()


@pytest.mark.parametrize("fmt", [0, 2, 3])
@pytest.mark.parametrize("block_size", [1, 7, 1000, 1250, 5000, 9999])
def test_oracle_lazer_class_refilters_the_first_chunk(oracle, pcq, fmt, block_size):
    """lazer.rs:80-116 never clears its buffer: chunk k filters points [0, points_in_chunk(k)) of chunk 0."""
    n = 5000 if block_size > 1 else 200
    image = oracle.synth_image(_spec(pcq, n, fmt), transposed=True)
    h, world, cls, rgb = _columns(oracle, image)
    lazer = oracle.lazer_from_last(image, block_size)
    target = int(np.bincount(cls).argmax())
    idx = []
    for k in range((n + block_size - 1) // block_size):
        in_chunk = min(block_size, n - k * block_size)
        idx.append(np.flatnonzero(cls[:in_chunk] == target))
    idx = np.concatenate(idx)
    bc = oracle.buffer_collector()
    assert oracle.search_lazer_class(lazer, target, bc) == OK
    got = bc.points()
    bc.free()
    assert got.tobytes() == _expect(world, cls, rgb, idx).tobytes()
    if block_size < n:
        assert len(got) != int((cls == target).sum()) or block_size == 1250  # not the true answer in general


def test_oracle_lazer_malformed_files(oracle, pcq):
    n = 2000
    image = oracle.synth_image(_spec(pcq, n, 2), transposed=True)
    h = oracle.parse_header(image[:400].tobytes())
    otp = h.offset_to_point_data
    good = oracle.lazer_from_last(image, 600)
    box = (list(h.min), list(h.max))

    def run(img, kind="bounds"):
        c = oracle.count_collector()
        rc = oracle.search_lazer_bounds(img, box[0], box[1], c) if kind == "bounds" else oracle.search_lazer_class(img, 2, c)
        cnt = c.point_count()
        c.free()
        return rc, cnt

    assert run(good) == (OK, n)
    # block size 0 -> division by zero panic (lazer_reader.rs:67)
    bad = good.copy()
    bad[otp:otp + 8] = 0
    assert run(bad)[0] == ERR_PANIC and run(bad, "class")[0] == ERR_PANIC
    # zero points -> `block_offsets[0]` on an empty Vec panics (lazer_reader.rs:123,143)
    bad = good.copy()
    bad[107:111] = 0
    assert run(bad)[0] == ERR_PANIC
    # file cut inside the last block
    assert run(good[:-50])[0] in (ERR_EOF, ERR_HEADER)
    assert run(good[:-50], "class")[0] in (ERR_EOF, ERR_HEADER)
    # cut inside the block table
    assert run(good[:otp + 12])[0] == ERR_EOF
    # the early-out comes after the constructor: a broken file and a far-away box still fails
    c = oracle.count_collector()
    far = [v + 1e6 for v in h.max]
    assert oracle.search_lazer_bounds(bad, far, far, c) == ERR_PANIC
    c.free()


def test_golden_lazer_file(oracle):
    """tests/golden/tiny_fmt2.lazer: blobs written by the real liblz4 (make_golden.py), answers by hand-checkable numpy."""
    with open(os.path.join(GOLDEN, "expected.json")) as f:
        exp = json.load(f)["lazer"]
    image = np.fromfile(os.path.join(GOLDEN, "tiny_fmt2.lazer"), dtype=np.uint8)
    for q in exp["bounds"]:
        c = oracle.buffer_collector()
        assert oracle.search_lazer_bounds(image, q["min"], q["max"], c) == OK
        pts = c.points()
        c.free()
        assert len(pts) == q["count"]
        assert pts.tobytes().hex() == q["points_hex"]
    for q in exp["class"]:
        c = oracle.count_collector()
        assert oracle.search_lazer_class(image, q["class"], c) == OK
        assert c.point_count() == q["count"]
        c.free()


# ---- differential fuzz: mutated frames through the real reader, the oracle and the product ---------------------
@needs_liblz4
def test_mutated_frames_differential(oracle, product):
    """Random damage (bit flips, truncation, splices, length-field edits) to real liblz4 frames: both
    readers must end exactly like lz4::Decoder over the real LZ4F_decompress — same error class, and the
    same bytes when the read succeeds — for the reference's read sizes."""
    rng = np.random.default_rng(0xF022)
    base = []
    text = contents()["text"][:140_000]
    xyz = contents()["xyz"][:100_000]
    for data in (text, xyz, b"abcabcabc" * 30, bytes(70_000)):
        for kw in (dict(), dict(independent=True, block_checksum=True, content_size=True), dict(content_checksum=False, block_checksum=True),
                   dict(content_checksum=False, content_size=True)):
            base.append((data, REAL.compress_frame(data, 4, **kw)))
    checked = errors = 0
    for it in range(400):
        data, frame = base[it % len(base)]
        fr = bytearray(frame)
        kind = it % 5
        if kind == 0:  # bit flips anywhere
            for _ in range(int(rng.integers(1, 4))):
                fr[int(rng.integers(0, len(fr)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:  # truncation
            fr = fr[:int(rng.integers(0, len(fr)))]
        elif kind == 2:  # damage in the header / first block header
            fr[int(rng.integers(4, min(24, len(fr))))] = int(rng.integers(0, 256))
        elif kind == 3:  # cut a slice out of the middle
            a = int(rng.integers(7, len(fr) - 1))
            b = min(len(fr), a + int(rng.integers(1, 64)))
            fr = fr[:a] + fr[b:]
        else:  # truncation exactly at a block boundary (walk the block headers of the intact frame)
            p = 4 + 2 + (8 if frame[4] & 8 else 0) + 1
            cuts = []
            while p + 4 <= len(frame):
                w = int.from_bytes(frame[p:p + 4], "little")
                if w == 0:
                    cuts += [p, p + 4]
                    break
                p += 4 + (w & 0x7FFFFFFF) + (4 if frame[4] & 0x10 else 0)
                cuts.append(p)
            fr = fr[:cuts[int(rng.integers(0, len(cuts)))]]
        fr = bytes(fr)
        need = [len(data), len(data) // 2 + 1, 65536, 65537, 1][int(rng.integers(0, 5))]
        need = min(need, len(data))
        for unit in (4, 1) if it % 3 else (2, 0):
            try:
                want, real = REAL.read_exact(fr, need, unit), OK
            except _lz4ref.UnexpectedEof:
                want, real = None, ERR_EOF
            except _lz4ref.LZ4Error:
                want, real = None, ERR_HEADER
            got_o, rc_o = oracle.lz4f_decode(fr, need, unit)
            got_p, rc_p = product(fr, need, unit)
            assert rc_o == real and rc_p == real, (it, kind, need, unit, real, rc_o, rc_p)
            if real == OK:
                assert got_o == want and got_p == want, (it, kind, need, unit)
            else:
                errors += 1
            checked += 1
    assert checked >= 800 and errors > 200


@needs_liblz4
def test_stored_blocks_are_passed_through_as_they_arrive(oracle, product):
    """Incompressible columns (noisy i32 positions) make liblz4 emit stored blocks.  LZ4F hands their bytes
    out as they arrive: a cut-off stored block still yields what is there, and its block checksum is looked at
    only by the read after its last byte."""
    data = contents()["random"][:66_000]
    frame = REAL.compress_frame(data, 4, False, False, True, False)  # block checksums, no content checksum
    first = int.from_bytes(frame[7:11], "little")
    assert first >> 31 and first & 0x7FFFFFFF == 65536  # a full stored block, then a short one

    def outcome(fr, need):
        res = set()
        for unit in (4, 2, 1):
            try:
                want, real = REAL.read_exact(fr, need, unit), OK
            except _lz4ref.UnexpectedEof:
                want, real = None, ERR_EOF
            except _lz4ref.LZ4Error:
                want, real = None, ERR_HEADER
            got_o, rc_o = oracle.lz4f_decode(fr, need, unit)
            got_p, rc_p = product(fr, need, unit)
            assert rc_o == real and rc_p == real, (need, unit, real, rc_o, rc_p)
            if real == OK:
                assert got_o == want == got_p
            res.add(real)
        assert len(res) == 1
        return res.pop()

    assert outcome(frame, 66_000) == OK
    cut = frame[:11 + 40_000]  # inside the first stored block
    assert outcome(cut, 40_000) == OK
    assert outcome(cut, 39_999) == OK
    assert outcome(cut, 40_001) == ERR_EOF
    bad = bytearray(frame)
    bad[11 + 65536] ^= 0xFF  # the first block's checksum
    bad = bytes(bad)
    assert outcome(bad, 65_536) == OK  # all of the block's data, checksum not looked at yet
    assert outcome(bad, 65_537) == ERR_HEADER
    flip = bytearray(frame)
    flip[11 + 100] ^= 1  # damaged data, intact checksum field
    assert outcome(bytes(flip), 65_536) == OK
    assert outcome(bytes(flip), 65_540) == ERR_HEADER


def test_lz4_reader_under_address_sanitizer(oracle, tmp_path):
    """The product's LZ4 Frame reader parses untrusted bytes on the host: run it, built with ASan + UBSan
    (CPU build only; GPU sanitizers are not available), over damaged frames of every kind."""
    import shutil
    import struct
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "lz4_asan")
    host = os.path.join(PKG, "host")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I" + os.path.join(ROOT, "include"), "-I" + host, os.path.join(ROOT, "tests", "native", "lz4_asan_driver.cpp"),
           os.path.join(host, "lz4_frame.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr and "cannot find" in r.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr
    rng = np.random.default_rng(99)
    c = contents()
    frames = []
    for data in (c["text"][:140_000], c["xyz"][:80_000], c["random"][:66_000], b"abcabcabc" * 30, bytes(70_000), c["period3"][:131_077]):
        for flags in (0, 1 | 2 | 8, 4, 2 | 4, 16 | 2, 4 | 8):
            frames.append((len(data), oracle.lz4f_compress(data, flags, 4)))
    corpus = str(tmp_path / "corpus.bin")
    n_cases = 0
    with open(corpus, "wb") as f:
        for it in range(3000):
            size, frame = frames[int(rng.integers(0, len(frames)))]
            fr = bytearray(frame)
            k = int(rng.integers(0, 6))
            if k == 0:
                for _ in range(int(rng.integers(1, 5))):
                    fr[int(rng.integers(0, len(fr)))] ^= 1 << int(rng.integers(0, 8))
            elif k == 1:
                fr = fr[:int(rng.integers(0, len(fr) + 1))]
            elif k == 2:
                fr[int(rng.integers(4, min(24, len(fr))))] = int(rng.integers(0, 256))
            elif k == 3:
                a = int(rng.integers(7, len(fr) - 1))
                fr = fr[:a] + fr[min(len(fr), a + int(rng.integers(1, 64))):]
            elif k == 4:  # random bytes spliced in (long literal / match length runs of 0xFF among them)
                a = int(rng.integers(7, len(fr) - 1))
                junk = bytes([255] * int(rng.integers(1, 40))) if it % 2 else rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8).tobytes()
                fr = fr[:a] + junk + fr[a:]
            need = int([size, size // 2 + 1, 65536, 65537, 1, 0, size + 5, 2 * size][int(rng.integers(0, 8))])
            unit = int([4, 2, 1, 0][int(rng.integers(0, 4))])
            f.write(struct.pack("<III", len(fr), need, unit) + bytes(fr))
            n_cases += 1
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, corpus], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    tag, cases, errors = r.stdout.split()
    assert tag == "ok" and int(cases) == n_cases and int(errors) > n_cases // 3


def test_host_file_parsers_under_address_sanitizer(oracle, pcq, tmp_path):
    """The host layer parses untrusted LAS headers and LAZER block / attribute tables before any GPU work.
    ASan + UBSan build of those parsers over mutated files; the error class must equal the oracle's."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    host = os.path.join(PKG, "host")
    exe = str(tmp_path / "host_parse_asan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-pthread",
           "-I" + os.path.join(ROOT, "include"), "-I" + host, os.path.join(ROOT, "tests", "native", "host_parse_asan_driver.cpp"),
           os.path.join(host, "core.cpp"), os.path.join(host, "search.cpp"), os.path.join(host, "lazer.cpp"),
           os.path.join(host, "lz4_frame.cpp"), "-L" + PKG, "-lpcq", "-Wl,-rpath," + PKG, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr and "cannot find" in r.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-3000:]
    rng = np.random.default_rng(77)
    image = oracle.synth_image(_spec(pcq, 3000, 2), transposed=True)
    good = oracle.lazer_from_last(image, 700, 4, 4)
    h = oracle.parse_header(good[:400].tobytes())
    otp = h.offset_to_point_data
    first_block = int(good[otp + 8: otp + 16].view("<u8")[0])
    paths = []
    for it in range(400):
        b = good.copy()
        k = it % 5
        if k == 0:  # anywhere in the LAS header
            b[int(rng.integers(0, 227))] = int(rng.integers(0, 256))
        elif k == 1:  # block size / block offsets table
            b[int(rng.integers(otp, otp + 8 + 8 * 5))] = int(rng.integers(0, 256))
        elif k == 2:  # attribute table of the first block
            b[int(rng.integers(first_block, first_block + 72))] = int(rng.integers(0, 256))
        elif k == 3:  # truncation somewhere in the tables
            b = b[:int(rng.integers(0, first_block + 100))]
        else:  # a 64-bit field set to an extreme
            at = int(rng.choice([otp, otp + 8, otp + 16, first_block, first_block + 8, first_block + 64]))
            extremes = [0, 1, 2 ** 63, 2 ** 64 - 1, len(good), len(good) + 1]
            b[at:at + 8] = np.frombuffer(extremes[int(rng.integers(0, len(extremes)))].to_bytes(8, "little"), np.uint8)
        p = str(tmp_path / f"m{it}.lazer")
        np.asarray(b).tofile(p)
        paths.append(p)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe] + paths, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-4000:])
    lines = r.stdout.splitlines()
    assert len(lines) == len(paths)
    failures = 0
    for line, p in zip(lines, paths):
        _, hrc, lrc = line.rsplit(" ", 2)
        img = np.fromfile(p, dtype=np.uint8)
        mn, mx = (C.c_double * 3)(), (C.c_double * 3)()
        want = oracle.lib.pcqo_lazer_mem_bounds(img.ctypes.data_as(C.c_void_p), img.size, mn, mx) if img.size else None
        if want is not None:
            assert int(lrc) == want, (p, lrc, want, oracle.err())
        failures += int(lrc) != 0
    assert 50 < failures < len(paths)
