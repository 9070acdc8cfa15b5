"""ctypes view of the image's liblz4.so.1 (1.9.x) — the same library lz4-sys (under the reference's
`lz4` crate, Cargo.lock) wraps.  TEST-SIDE ONLY: it pins the oracle's and the product's LZ4 Frame
readers against the real implementation; the product never loads it.

`read_exact` re-enacts lz4 1.23.2's `impl Read for Decoder` loop (decoder.rs: 32 KiB input buffer,
`next` hint, LZ4F_decompress until the destination is full) followed by std's read_exact, i.e. what
readers/src/lazer_reader.rs:598-600 does to a column blob.
"""
import ctypes as C
import ctypes.util


class FrameInfo(C.Structure):
    _fields_ = [("blockSizeID", C.c_int), ("blockMode", C.c_int), ("contentChecksumFlag", C.c_int),
                ("frameType", C.c_int), ("contentSize", C.c_ulonglong), ("dictID", C.c_uint),
                ("blockChecksumFlag", C.c_int)]


class Preferences(C.Structure):
    _fields_ = [("frameInfo", FrameInfo), ("compressionLevel", C.c_int), ("autoFlush", C.c_uint),
                ("favorDecSpeed", C.c_uint), ("reserved", C.c_uint * 3)]


class LZ4Error(Exception):
    pass


class UnexpectedEof(Exception):
    pass


def load():
    for name in ("liblz4.so.1", ctypes.util.find_library("lz4")):
        if not name:
            continue
        try:
            return RealLZ4(C.CDLL(name))
        except OSError:
            continue
    return None


class RealLZ4:
    def __init__(self, lib):
        self.lib = lib
        lib.LZ4F_compressFrameBound.restype = C.c_size_t
        lib.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.POINTER(Preferences)]
        lib.LZ4F_compressFrame.restype = C.c_size_t
        lib.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(Preferences)]
        lib.LZ4F_isError.argtypes = [C.c_size_t]
        lib.LZ4F_getErrorName.restype = C.c_char_p
        lib.LZ4F_getErrorName.argtypes = [C.c_size_t]
        lib.LZ4F_createDecompressionContext.restype = C.c_size_t
        lib.LZ4F_createDecompressionContext.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        lib.LZ4F_freeDecompressionContext.argtypes = [C.c_void_p]
        lib.LZ4F_decompress.restype = C.c_size_t
        lib.LZ4F_decompress.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p,
                                        C.POINTER(C.c_size_t), C.c_void_p]
        lib.LZ4_versionNumber.restype = C.c_int

    def version(self):
        return self.lib.LZ4_versionNumber()

    def compress_frame(self, data: bytes, block_id=0, independent=False, content_checksum=True, block_checksum=False,
                       content_size=False, level=0) -> bytes:
        p = Preferences()
        p.frameInfo.blockSizeID = block_id
        p.frameInfo.blockMode = 1 if independent else 0
        p.frameInfo.contentChecksumFlag = 1 if content_checksum else 0
        p.frameInfo.blockChecksumFlag = 1 if block_checksum else 0
        p.frameInfo.contentSize = len(data) if content_size else 0
        p.compressionLevel = level
        cap = self.lib.LZ4F_compressFrameBound(len(data), C.byref(p))
        buf = C.create_string_buffer(cap)
        n = self.lib.LZ4F_compressFrame(buf, cap, data, len(data), C.byref(p))
        if self.lib.LZ4F_isError(n):
            raise LZ4Error(self.lib.LZ4F_getErrorName(n).decode())
        return buf.raw[:n]

    def read_exact(self, frame: bytes, need: int, unit: int = 0) -> bytes:
        """`need` bytes out of lz4::Decoder::new(Cursor::new(frame)) with std::io::Read::read_exact, `unit`
        bytes per call (4 = read_i32, 1 = read_u8, 2 = read_u16; 0 = everything in one call)."""
        ctx = C.c_void_p()
        rc = self.lib.LZ4F_createDecompressionContext(C.byref(ctx), 100)
        assert not self.lib.LZ4F_isError(rc)
        try:
            src = C.create_string_buffer(frame, len(frame)) if frame else C.create_string_buffer(1)
            out = C.create_string_buffer(max(need, 1))
            state = {"pos": 0, "len": 0, "next": 11, "cursor": 0, "base": 0}

            def read(dst_off, want):  # Decoder::read
                if state["next"] == 0 or want == 0:
                    return 0
                got = 0
                while got == 0:
                    if state["pos"] >= state["len"]:
                        take = min(32 * 1024, state["next"], len(frame) - state["cursor"])
                        state["base"] = state["cursor"]
                        state["len"] = take
                        state["cursor"] += take
                        if take == 0:
                            break
                        state["pos"] = 0
                        state["next"] -= take
                    while got < want and state["pos"] < state["len"]:
                        s_sz = C.c_size_t(state["len"] - state["pos"])
                        d_sz = C.c_size_t(want - got)
                        ret = self.lib.LZ4F_decompress(ctx, C.byref(out, dst_off + got), C.byref(d_sz),
                                                       C.byref(src, state["base"] + state["pos"]), C.byref(s_sz), None)
                        if self.lib.LZ4F_isError(ret):
                            raise LZ4Error(self.lib.LZ4F_getErrorName(ret).decode())
                        state["pos"] += s_sz.value
                        got += d_sz.value
                        if ret == 0:
                            state["next"] = 0
                            return got
                        if state["next"] < ret:
                            state["next"] = ret
                return got

            done = 0
            while done < need:
                stop = need if unit <= 0 else min(need, done + unit)
                while done < stop:  # read_exact
                    k = read(done, stop - done)
                    if k == 0:
                        raise UnexpectedEof("failed to fill whole buffer")
                    done += k
            return out.raw[:need]
        finally:
            self.lib.LZ4F_freeDecompressionContext(ctx)
