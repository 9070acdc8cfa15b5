"""ctypes view of oracle/liboracle.so — the CPU restatement of the reference's algorithm.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, as the checker / reported baseline.
"""
from __future__ import annotations

import ctypes as C
import importlib
import os
from typing import Sequence

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_pkg = importlib.import_module("adhoc-queries-pointclouds_amd")
SynthSpec = _pkg.SynthSpec
POINT_DTYPE = _pkg.POINT_DTYPE

QUERY_BOUNDS, QUERY_CLASS = 0, 1
OK, ERR_IO, ERR_HEADER, ERR_FORMAT, ERR_EXTENSION, ERR_EOF, ERR_GRID, ERR_PANIC, ERR_ARG = 0, -1, -2, -3, -4, -5, -6, -7, -8


class LasHeader(C.Structure):
    _fields_ = [("version_major", C.c_uint8), ("version_minor", C.c_uint8), ("header_size", C.c_uint16),
                ("offset_to_point_data", C.c_uint32), ("number_of_vlrs", C.c_uint32),
                ("point_data_record_format", C.c_uint8), ("point_data_record_length", C.c_uint16),
                ("legacy_number_of_points", C.c_uint32), ("large_number_of_points", C.c_uint64),
                ("scale", C.c_double * 3), ("offset", C.c_double * 3), ("min", C.c_double * 3), ("max", C.c_double * 3),
                ("number_of_points", C.c_uint64)]


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"oracle error {code}: {msg}")
        self.code = code
        self.message = msg


class OracleCollector:
    def __init__(self, o: "Oracle", handle):
        self.o, self.h = o, handle

    def collect_one(self, x, y, z, r=0, g=0, b=0, cls=0):
        p = np.zeros(1, dtype=POINT_DTYPE)
        p[0] = (x, y, z, r, g, b, cls)
        self.o.lib.pcqo_collector_collect_one(self.h, p.ctypes.data_as(C.c_void_p))

    def point_count(self) -> int:
        return self.o.lib.pcqo_collector_point_count(self.h)

    def has_points(self) -> bool:
        return bool(self.o.lib.pcqo_collector_has_points(self.h))

    def points(self) -> np.ndarray:
        n = self.o.lib.pcqo_collector_points(self.h, None, 0)
        out = np.zeros(n, dtype=POINT_DTYPE)
        if n:
            self.o.lib.pcqo_collector_points(self.h, out.ctypes.data_as(C.c_void_p), n)
        return out

    def grid_cells(self) -> np.ndarray:
        n = self.o.lib.pcqo_collector_grid_cells(self.h, None, 0)
        out = np.zeros(n, dtype=np.uint64)
        if n:
            self.o.lib.pcqo_collector_grid_cells(self.h, out.ctypes.data_as(C.c_void_p), n)
        return out

    def grid_params(self):
        dims, bits = (C.c_uint64 * 3)(), (C.c_uint64 * 3)()
        self.o.lib.pcqo_collector_grid_params(self.h, dims, bits)
        return list(dims), list(bits)

    def free(self):
        if self.h:
            self.o.lib.pcqo_collector_free(self.h)
            self.h = None


class Oracle:
    def __init__(self):
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        self.lib = lib = C.CDLL(path)
        vp, u64 = C.c_void_p, C.c_uint64
        P = C.POINTER
        dd = P(C.c_double)
        lib.pcqo_last_error.restype = C.c_char_p
        lib.pcqo_f64_as_i64.restype = C.c_int64
        lib.pcqo_f64_as_i64.argtypes = [C.c_double]
        lib.pcqo_f64_as_u64.restype = C.c_uint64
        lib.pcqo_f64_as_u64.argtypes = [C.c_double]
        lib.pcqo_parse_las_header.argtypes = [vp, C.c_size_t, C.c_int, P(LasHeader)]
        lib.pcqo_box_to_local.argtypes = [dd, dd, dd, dd, P(C.c_int64), P(C.c_int64)]
        lib.pcqo_aabb_intersects.argtypes = [dd, dd, dd, dd]
        for name in ("pcqo_collector_new_count", "pcqo_collector_new_buffer"):
            getattr(lib, name).restype = vp
        lib.pcqo_collector_new_grid.restype = vp
        lib.pcqo_collector_new_grid.argtypes = [dd, dd, C.c_double]
        lib.pcqo_collector_free.argtypes = [vp]
        lib.pcqo_collector_collect_one.argtypes = [vp, vp]
        lib.pcqo_collector_point_count.restype = u64
        lib.pcqo_collector_point_count.argtypes = [vp]
        lib.pcqo_collector_has_points.argtypes = [vp]
        lib.pcqo_collector_points.restype = u64
        lib.pcqo_collector_points.argtypes = [vp, vp, u64]
        lib.pcqo_collector_grid_cells.restype = u64
        lib.pcqo_collector_grid_cells.argtypes = [vp, vp, u64]
        lib.pcqo_collector_grid_params.argtypes = [vp, P(u64), P(u64)]
        lib.pcqo_search_last_mem_by_bounds_optimized.argtypes = [vp, C.c_size_t, dd, dd, vp]
        lib.pcqo_search_last_mem_by_classification_optimized.argtypes = [vp, C.c_size_t, C.c_uint8, vp]
        lib.pcqo_search_las_mem_by_bounds_optimized.argtypes = [vp, C.c_size_t, dd, dd, vp, P(C.c_int)]
        lib.pcqo_search_las_mem_by_classification_optimized.argtypes = [vp, C.c_size_t, C.c_uint8, vp]
        lib.pcqo_search_file.argtypes = [C.c_char_p, C.c_int, dd, dd, C.c_uint8, vp, P(C.c_int)]
        lib.pcqo_count_files_parallel.argtypes = [P(vp), P(C.c_size_t), C.c_size_t, C.c_int, dd, dd, C.c_uint8, C.c_int, P(u64)]
        lib.pcqo_synth_mix.restype = u64
        lib.pcqo_synth_mix.argtypes = [u64, u64]
        lib.pcqo_synth_fill_columns.argtypes = [P(SynthSpec), u64, u64, vp, vp]
        lib.pcqo_synth_image_size.restype = C.c_size_t
        lib.pcqo_synth_image_size.argtypes = [P(SynthSpec)]
        lib.pcqo_synth_build_image.argtypes = [P(SynthSpec), C.c_int, vp, C.c_size_t, C.c_int]
        lib.pcqo_synth_write_file.argtypes = [P(SynthSpec), C.c_int, C.c_char_p, C.c_int]
        lib.pcqo_synth_build_header.argtypes = [P(SynthSpec), vp]
        lib.pcqo_search_lazer_mem_by_bounds.argtypes = [vp, C.c_size_t, dd, dd, vp]
        lib.pcqo_search_lazer_mem_by_classification.argtypes = [vp, C.c_size_t, C.c_uint8, vp]
        lib.pcqo_lazer_mem_bounds.argtypes = [vp, C.c_size_t, dd, dd]
        lib.pcqo_lz4f_decode.restype = C.c_int64
        lib.pcqo_lz4f_decode.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.c_size_t]
        lib.pcqo_xxh32.restype = C.c_uint32
        lib.pcqo_xxh32.argtypes = [vp, C.c_size_t]
        lib.pcqo_lz4f_compress.restype = vp
        lib.pcqo_lz4f_compress.argtypes = [vp, C.c_size_t, C.c_uint, C.c_int, P(C.c_size_t)]
        lib.pcqo_lazer_from_last.restype = vp
        lib.pcqo_lazer_from_last.argtypes = [vp, C.c_size_t, u64, C.c_uint, C.c_int, P(C.c_size_t)]
        lib.pcqo_free.argtypes = [vp]

    # --- helpers ------------------------------------------------------------------------------
    def err(self) -> str:
        return self.lib.pcqo_last_error().decode("utf-8", "replace")

    def check(self, rc):
        if rc != OK:
            raise OracleError(rc, self.err())

    def f64_as_i64(self, v):
        return self.lib.pcqo_f64_as_i64(v)

    def f64_as_u64(self, v):
        return self.lib.pcqo_f64_as_u64(v)

    def parse_header(self, data: bytes, mask_format=False) -> LasHeader:
        h = LasHeader()
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        self.check(self.lib.pcqo_parse_las_header(buf, len(data), int(mask_format), C.byref(h)))
        return h

    def box_to_local(self, bmin, bmax, scale, offset):
        lmin, lmax = (C.c_int64 * 3)(), (C.c_int64 * 3)()
        self.check(self.lib.pcqo_box_to_local(_d3(bmin), _d3(bmax), _d3(scale), _d3(offset), lmin, lmax))
        return list(lmin), list(lmax)

    def aabb_intersects(self, amin, amax, bmin, bmax) -> bool:
        return bool(self.lib.pcqo_aabb_intersects(_d3(amin), _d3(amax), _d3(bmin), _d3(bmax)))

    # --- collectors ---------------------------------------------------------------------------
    def count_collector(self):
        return OracleCollector(self, self.lib.pcqo_collector_new_count())

    def buffer_collector(self):
        return OracleCollector(self, self.lib.pcqo_collector_new_buffer())

    def grid_collector(self, bmin, bmax, cell):
        h = self.lib.pcqo_collector_new_grid(_d3(bmin), _d3(bmax), float(cell))
        if not h:
            raise OracleError(ERR_GRID, self.err())
        return OracleCollector(self, h)

    # --- scans on memory images ---------------------------------------------------------------
    @staticmethod
    def _img(image):
        a = np.frombuffer(image, dtype=np.uint8) if not isinstance(image, np.ndarray) else image
        return a, a.ctypes.data_as(C.c_void_p), a.size

    def search_last_bounds(self, image, bmin, bmax, coll: OracleCollector) -> int:
        a, p, n = self._img(image)
        return self.lib.pcqo_search_last_mem_by_bounds_optimized(p, n, _d3(bmin), _d3(bmax), coll.h)

    def search_last_class(self, image, cls, coll: OracleCollector) -> int:
        a, p, n = self._img(image)
        return self.lib.pcqo_search_last_mem_by_classification_optimized(p, n, cls, coll.h)

    def search_las_bounds(self, image, bmin, bmax, coll: OracleCollector):
        a, p, n = self._img(image)
        rec = C.c_int(-1)
        rc = self.lib.pcqo_search_las_mem_by_bounds_optimized(p, n, _d3(bmin), _d3(bmax), coll.h, C.byref(rec))
        return rc, rec.value

    def search_las_class(self, image, cls, coll: OracleCollector) -> int:
        a, p, n = self._img(image)
        return self.lib.pcqo_search_las_mem_by_classification_optimized(p, n, cls, coll.h)

    # --- LAZER / LZ4 ----------------------------------------------------------------------------
    def search_lazer_bounds(self, image, bmin, bmax, coll: OracleCollector) -> int:
        a, p, n = self._img(image)
        return self.lib.pcqo_search_lazer_mem_by_bounds(p, n, _d3(bmin), _d3(bmax), coll.h)

    def search_lazer_class(self, image, cls, coll: OracleCollector) -> int:
        a, p, n = self._img(image)
        return self.lib.pcqo_search_lazer_mem_by_classification(p, n, cls, coll.h)

    def xxh32(self, data: bytes) -> int:
        a, p, n = self._img(bytes(data) or b"\0")
        return self.lib.pcqo_xxh32(p, len(data))

    def lz4f_decode(self, frame: bytes, need: int, unit: int = 0):
        """(bytes, 0) or (None, error code): the first `need` bytes of the frame, `unit` bytes per read."""
        a, p, n = self._img(bytes(frame) or b"\0")
        out = np.zeros(max(need, 1), dtype=np.uint8)
        rc = self.lib.pcqo_lz4f_decode(p, len(frame), out.ctypes.data_as(C.c_void_p), need, unit)
        return (out[:need].tobytes(), 0) if rc >= 0 else (None, int(rc))

    def _take(self, ptr, n) -> bytes:
        if not ptr:
            raise OracleError(ERR_ARG, "oracle writer failed")
        data = C.string_at(ptr, n)
        self.lib.pcqo_free(ptr)
        return data

    def lz4f_compress(self, content: bytes, flags=4, block_id=4) -> bytes:
        a, p, n = self._img(bytes(content) or b"\0")
        out_n = C.c_size_t(0)
        return self._take(self.lib.pcqo_lz4f_compress(p, len(content), flags, block_id, C.byref(out_n)), out_n.value)

    def lazer_from_last(self, last_image, block_size: int, flags=4, block_id=4) -> np.ndarray:
        a, p, n = self._img(last_image)
        out_n = C.c_size_t(0)
        raw = self._take(self.lib.pcqo_lazer_from_last(p, n, block_size, flags, block_id, C.byref(out_n)), out_n.value)
        return np.frombuffer(raw, dtype=np.uint8).copy()

    def search_file(self, path, kind, bmin, bmax, cls, coll: OracleCollector):
        rec = C.c_int(-1)
        rc = self.lib.pcqo_search_file(path.encode(), kind, _d3(bmin or (0, 0, 0)), _d3(bmax or (0, 0, 0)), cls, coll.h,
                                       C.byref(rec))
        return rc, rec.value

    def count_files_parallel(self, images: Sequence[np.ndarray], kind, bmin, bmax, cls, threads) -> int:
        n = len(images)
        ptrs = (C.c_void_p * n)(*[im.ctypes.data for im in images])
        lens = (C.c_size_t * n)(*[im.size for im in images])
        total = C.c_uint64(0)
        self.check(self.lib.pcqo_count_files_parallel(ptrs, lens, n, kind, _d3(bmin or (0, 0, 0)), _d3(bmax or (0, 0, 0)),
                                                      cls, threads, C.byref(total)))
        return total.value

    # --- synthetic data -----------------------------------------------------------------------
    def synth_columns(self, spec: SynthSpec, first=0, count=None):
        count = spec.n - first if count is None else count
        xyz = np.empty(count * 3, dtype=np.int32)
        cls = np.empty(count, dtype=np.uint8)
        self.lib.pcqo_synth_fill_columns(C.byref(spec), first, count, xyz.ctypes.data_as(C.c_void_p),
                                         cls.ctypes.data_as(C.c_void_p))
        return xyz.reshape(-1, 3), cls

    def synth_image(self, spec: SynthSpec, transposed=True, threads=4) -> np.ndarray:
        size = self.lib.pcqo_synth_image_size(C.byref(spec))
        out = np.empty(size, dtype=np.uint8)
        self.check(self.lib.pcqo_synth_build_image(C.byref(spec), int(transposed), out.ctypes.data_as(C.c_void_p), size, threads))
        return out

    def synth_write(self, spec: SynthSpec, path: str, transposed=None, threads=4):
        if transposed is None:
            transposed = path.endswith(".last")
        self.check(self.lib.pcqo_synth_write_file(C.byref(spec), int(transposed), path.encode(), threads))

    def synth_header(self, spec: SynthSpec) -> bytes:
        out = (C.c_uint8 * 227)()
        self.check(self.lib.pcqo_synth_build_header(C.byref(spec), out))
        return bytes(out)


def canon_points(pts: np.ndarray) -> np.ndarray:
    """Order-independent canonical form of a point set (the reference's grid output order is a
    HashMap iteration order): sort by the raw 31-byte records."""
    raw = np.ascontiguousarray(pts).view(np.uint8).reshape(-1, 31)
    order = np.lexsort(raw.T[::-1])
    return pts[order]
