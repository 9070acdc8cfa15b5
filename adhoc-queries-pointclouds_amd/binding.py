"""ctypes view of include/pcq.h and include/pcq_synth.h (libpcq.so).

Plumbing only: every scan runs in the HIP library.  Loading fails loudly when the library has not
been built (``python -c "import __graft_entry__ as g; g.build()"``).
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)

PCQ_OK = 0
PCQ_ERR_IO, PCQ_ERR_HEADER, PCQ_ERR_FORMAT, PCQ_ERR_EXTENSION, PCQ_ERR_EOF = -1, -2, -3, -4, -5
PCQ_ERR_GRID, PCQ_ERR_PANIC, PCQ_ERR_ARG, PCQ_ERR_HIP, PCQ_ERR_CAPACITY = -6, -7, -8, -9, -10
PCQ_ERR_UNSUPPORTED, PCQ_ERR_NOMEM = -11, -12
PRED_BOUNDS, PRED_CLASS, PRED_BOUNDS_F64 = 0, 1, 2

# readers/src/lib.rs:10-19 — packed 31-byte result record
POINT_DTYPE = np.dtype(
    [("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("r", "<u2"), ("g", "<u2"), ("b", "<u2"), ("classification", "u1")]
)
assert POINT_DTYPE.itemsize == 31


class PcqError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"pcq error {code}: {message}")
        self.code = code
        self.message = message


class Point(C.Structure):
    _pack_ = 1
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double), ("r", C.c_uint16), ("g", C.c_uint16),
                ("b", C.c_uint16), ("classification", C.c_uint8)]


class Columns(C.Structure):
    _fields_ = [("xyz", C.c_void_p), ("cls", C.c_void_p), ("rgb", C.c_void_p), ("xyz_stride", C.c_uint64),
                ("cls_stride", C.c_uint64), ("rgb_stride", C.c_uint64), ("n", C.c_uint64), ("first_index", C.c_uint64),
                ("scale", C.c_double * 3), ("offset", C.c_double * 3)]


class Predicate(C.Structure):
    _fields_ = [("kind", C.c_int32), ("cls", C.c_uint8), ("_pad", C.c_uint8 * 3), ("lmin", C.c_int64 * 3),
                ("lmax", C.c_int64 * 3), ("wmin", C.c_double * 3), ("wmax", C.c_double * 3)]

    @staticmethod
    def bounds_f64(wmin: Sequence[float], wmax: Sequence[float]) -> "Predicate":
        p = Predicate()
        p.kind = PRED_BOUNDS_F64
        for a in range(3):
            p.wmin[a] = float(wmin[a])
            p.wmax[a] = float(wmax[a])
        return p

    @staticmethod
    def bounds(lmin: Sequence[int], lmax: Sequence[int]) -> "Predicate":
        p = Predicate()
        p.kind = PRED_BOUNDS
        for a in range(3):
            p.lmin[a] = int(lmin[a])
            p.lmax[a] = int(lmax[a])
        return p

    @staticmethod
    def classification(cls: int) -> "Predicate":
        p = Predicate()
        p.kind = PRED_CLASS
        p.cls = int(cls)
        return p


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("gcn_arch", C.c_char * 64), ("compute_units", C.c_int),
                ("wavefront_size", C.c_int), ("hbm_bytes", C.c_uint64), ("lds_bytes_per_block", C.c_uint64),
                ("clock_khz", C.c_int)]


class IndexStats(C.Structure):
    _fields_ = [("chunks", C.c_uint64), ("skipped", C.c_uint64), ("whole", C.c_uint64), ("scanned", C.c_uint64),
                ("built", C.c_uint64)]


class SynthSpec(C.Structure):
    """include/pcq_synth.h: pcq_synth_spec (the test-side host generator uses the same layout)."""
    _fields_ = [("seed", C.c_uint64), ("n", C.c_uint64), ("format", C.c_uint32), ("n_classes", C.c_uint32),
                ("scale", C.c_double * 3), ("offset", C.c_double * 3), ("lo", C.c_int32 * 3), ("span", C.c_uint32 * 3),
                ("zo_prob16", C.c_uint32), ("zo_lo", C.c_int32), ("zo_span", C.c_uint32), ("cls_cum16", C.c_uint32 * 8),
                ("cls_val", C.c_uint8 * 8)]


def lib_path() -> str:
    """libpcq.so — or, for the measurement tools in tools/ (PCQ_LAB=1), libpcq_lab.so: the same sources plus the
    superseded kernel shapes and read microbenchmarks of csrc/lab/ (make -C csrc lab); PCQ_LAB=stamps: libpcq_stamps.so,
    the timing build of the grid folds (make -C csrc stamps)."""
    which = os.environ.get("PCQ_LAB")
    return os.path.join(_HERE, "libpcq_lab.so" if which == "1" else "libpcq_stamps.so" if which == "stamps" else "libpcq.so")


_lib = None


def _one_hip_runtime() -> Optional[str]:
    """One HIP runtime per process.  libpcq.so needs `libamdhip64.so.7`; a PyTorch wheel ships its own copy of that
    library (same SONAME, found through torch/lib's RUNPATH).  Two copies in one process means two HSA runtimes, and
    the one that initialises second reports "no ROCm-capable device".  The dynamic loader reuses an already mapped
    library with the SONAME a new one asks for, so: when PyTorch is installed, its copy is mapped first (without
    importing torch) and libpcq.so binds to it — whichever of the two is imported first afterwards, there is one
    runtime.  Without PyTorch (the `query` CLI, a Rust host) libpcq.so finds ROCm's through its RUNPATH.
    PCQ_HIP_RUNTIME=rocm keeps ROCm's runtime even when PyTorch is installed."""
    if os.environ.get("PCQ_HIP_RUNTIME", "") == "rocm":
        return None
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(cand):
        return None
    C.CDLL(cand, mode=C.RTLD_GLOBAL)
    return cand


_hip_runtime_path = None


def hip_runtime_path() -> Optional[str]:
    """The libamdhip64 mapped on behalf of libpcq.so by load_library() (None: ROCm's, through the RUNPATH)."""
    return _hip_runtime_path


def load_library() -> C.CDLL:
    """Loads libpcq.so; raises if it has not been built.  There is no fallback."""
    global _lib, _hip_runtime_path
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build the HIP library first (make -C {_HERE}/csrc, or "
                           "__graft_entry__.build()). This package has no CPU path.")
    _hip_runtime_path = _one_hip_runtime()
    lib = C.CDLL(path)
    vp, u64, i64 = C.c_void_p, C.c_uint64, C.c_int64
    P = C.POINTER
    sig = {
        "pcq_init": (C.c_int, [C.c_int, P(vp)]),
        "pcq_shutdown": (C.c_int, [vp]),
        "pcq_last_error": (C.c_char_p, []),
        "pcq_abi_version": (C.c_int, []),
        "pcq_get_device_info": (C.c_int, [vp, P(DeviceInfo)]),
        "pcq_ctx_stream": (vp, [vp]),
        "pcq_ctx_synchronize": (C.c_int, [vp]),
        "pcq_box_to_local": (C.c_int, [P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_double), P(i64), P(i64)]),
        "pcq_collector_new_count": (C.c_int, [vp, P(vp)]),
        "pcq_collector_new_count_at": (C.c_int, [vp, vp, P(vp)]),
        "pcq_collector_new_buffer": (C.c_int, [vp, P(vp)]),
        "pcq_collector_new_grid": (C.c_int, [vp, P(C.c_double), P(C.c_double), C.c_double, P(vp)]),
        "pcq_collector_free": (C.c_int, [vp]),
        "pcq_collector_point_count": (C.c_int, [vp, P(u64)]),
        "pcq_collector_has_points": (C.c_int, [vp]),
        "pcq_collector_points": (C.c_int, [vp, vp, u64, P(u64)]),
        "pcq_collector_grid_cells": (C.c_int, [vp, vp, u64, P(u64)]),
        "pcq_collector_grid_params": (C.c_int, [vp, P(u64), P(u64)]),
        "pcq_collector_reset": (C.c_int, [vp]),
        "pcq_collector_flush": (C.c_int, [vp]),
        "pcq_scan_dev": (C.c_int, [vp, P(Columns), P(Predicate), vp, vp]),
        "pcq_scan_host": (C.c_int, [vp, P(Columns), P(Predicate), vp]),
        "pcq_scan_fd": (C.c_int, [vp, C.c_int, P(Columns), P(Predicate), vp]),
        "pcq_scan_host_nowait": (C.c_int, [vp, P(Columns), P(Predicate), vp]),
        "pcq_scan_fd_nowait": (C.c_int, [vp, C.c_int, P(Columns), P(Predicate), vp]),
        "pcq_prepare_host_scans": (C.c_int, [vp]),
        "pcq_scan_dev_count_batch": (C.c_int, [vp, P(Columns), P(Predicate), C.c_size_t, vp, vp]),
        "pcq_allreduce_sum_u64": (C.c_int, [P(vp), P(vp), P(vp), C.c_int]),
        "pcq_allreduce_prepare": (C.c_int, [P(C.c_int), C.c_int]),
        "pcq_read_fd_to_device": (C.c_int, [vp, C.c_int, u64, u64, vp]),
        "pcq_index_new": (C.c_int, [vp, P(vp)]),
        "pcq_index_free": (C.c_int, [vp]),
        "pcq_index_get_stats": (C.c_int, [vp, P(IndexStats)]),
        "pcq_scan_dev_indexed": (C.c_int, [vp, P(Columns), P(Predicate), vp, vp, vp]),
        "pcq_device_alloc": (C.c_int, [vp, u64, P(vp)]),
        "pcq_device_free": (C.c_int, [vp, vp]),
        "pcq_copy_to_device": (C.c_int, [vp, vp, vp, u64]),
        "pcq_copy_to_host": (C.c_int, [vp, vp, vp, u64]),
        "pcq_device_memset": (C.c_int, [vp, vp, C.c_int, u64, vp]),
        "pcq_set_option": (C.c_int, [vp, C.c_char_p, i64]),
        "pcq_get_option": (C.c_int, [vp, C.c_char_p, P(i64)]),
        "pcq_bind_thread_near_device": (C.c_int, [vp]),
        "pcq_synth_fill_dev": (C.c_int, [vp, P(SynthSpec), u64, u64, vp, vp, vp]),
    }
    lab_sig = {  # include/pcq_lab.h: present in libpcq_lab.so only
        "pcq_membench_read": (C.c_int, [vp, vp, u64, C.c_int, C.c_int, C.c_int, vp]),
        "pcq_membench_read_tiles": (C.c_int, [vp, vp, u64, C.c_int, C.c_int, C.c_int, vp]),
        "pcq_membench_read_xcd": (C.c_int, [vp, vp, u64, C.c_int, C.c_int, C.c_int, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in lab_sig.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def declared_symbols(headers: Optional[Sequence[str]] = None) -> list:
    """Function names declared in include/*.h (the ABI contract)."""
    inc = os.path.join(_ROOT, "include")
    names = []
    for h in headers or sorted(os.listdir(inc)):
        if not h.endswith(".h"):
            continue
        text = open(os.path.join(inc, h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(pcq[a-z0-9_]*)\s*\(", text)
    seen, out = set(), []
    for n in names:
        if n not in seen:
            seen.add(n)
            out.append(n)
    return out


def exported_symbols(path: str) -> set:
    """Dynamic symbols a shared library exports (read with ctypes only: no GPU call)."""
    import subprocess
    txt = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1] for line in txt.splitlines() if line.strip()}


def _check(rc: int) -> None:
    if rc != PCQ_OK:
        raise PcqError(rc, load_library().pcq_last_error().decode("utf-8", "replace"))


def _d3(v) -> C.Array:
    return (C.c_double * 3)(*[float(x) for x in v])


def box_to_local(bmin, bmax, scale, offset):
    """last.rs:98-109 via the library (pure host arithmetic; needs no device)."""
    lib = load_library()
    lmin, lmax = (C.c_int64 * 3)(), (C.c_int64 * 3)()
    _check(lib.pcq_box_to_local(_d3(bmin), _d3(bmax), _d3(scale), _d3(offset), lmin, lmax))
    return list(lmin), list(lmax)


class Collector:
    def __init__(self, ctx: "Context", handle: int, kind: str):
        self.ctx, self.handle, self.kind = ctx, C.c_void_p(handle), kind

    def point_count(self) -> int:
        n = C.c_uint64(0)
        _check(self.ctx.lib.pcq_collector_point_count(self.handle, C.byref(n)))
        return n.value

    def has_points(self) -> bool:
        return bool(self.ctx.lib.pcq_collector_has_points(self.handle))

    def points(self) -> np.ndarray:
        n = C.c_uint64(0)
        _check(self.ctx.lib.pcq_collector_points(self.handle, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=POINT_DTYPE)
        if n.value:
            _check(self.ctx.lib.pcq_collector_points(self.handle, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out

    def grid_cells(self) -> np.ndarray:
        n = C.c_uint64(0)
        _check(self.ctx.lib.pcq_collector_grid_cells(self.handle, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint64)
        if n.value:
            _check(self.ctx.lib.pcq_collector_grid_cells(self.handle, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out

    def grid_params(self):
        dims, bits = (C.c_uint64 * 3)(), (C.c_uint64 * 3)()
        _check(self.ctx.lib.pcq_collector_grid_params(self.handle, dims, bits))
        return list(dims), list(bits)

    def reset(self) -> None:
        _check(self.ctx.lib.pcq_collector_reset(self.handle))

    def flush(self) -> None:
        _check(self.ctx.lib.pcq_collector_flush(self.handle))

    def free(self) -> None:
        if self.handle:
            self.ctx.lib.pcq_collector_free(self.handle)
            self.handle = C.c_void_p(None)


class Context:
    """One pcq_ctx (a GPU + its stream, staging and scratch)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        _check(self.lib.pcq_init(device, C.byref(h)))
        self.handle = h

    def close(self) -> None:
        if self.handle:
            self.lib.pcq_shutdown(self.handle)
            self.handle = C.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def device_info(self) -> dict:
        d = DeviceInfo()
        _check(self.lib.pcq_get_device_info(self.handle, C.byref(d)))
        return {"name": d.name.decode(), "gcn_arch": d.gcn_arch.decode(), "compute_units": d.compute_units,
                "wavefront_size": d.wavefront_size, "hbm_bytes": d.hbm_bytes,
                "lds_bytes_per_block": d.lds_bytes_per_block, "clock_khz": d.clock_khz}

    def set_option(self, key: str, value: int) -> None:
        _check(self.lib.pcq_set_option(self.handle, key.encode(), int(value)))

    def get_option(self, key: str) -> int:
        v = C.c_int64()
        _check(self.lib.pcq_get_option(self.handle, key.encode(), C.byref(v)))
        return v.value

    def synchronize(self) -> None:
        _check(self.lib.pcq_ctx_synchronize(self.handle))

    def stream_handle(self) -> int:
        """The context's own HIP stream (hipStream_t as an integer): what scans without a caller's stream and every fold run on."""
        return int(self.lib.pcq_ctx_stream(self.handle) or 0)

    # collectors -------------------------------------------------------------------------------
    def count_collector(self, device_counter: Optional[int] = None) -> Collector:
        h = C.c_void_p()
        if device_counter is None:
            _check(self.lib.pcq_collector_new_count(self.handle, C.byref(h)))
        else:
            _check(self.lib.pcq_collector_new_count_at(self.handle, C.c_void_p(device_counter), C.byref(h)))
        return Collector(self, h.value, "count")

    def buffer_collector(self) -> Collector:
        h = C.c_void_p()
        _check(self.lib.pcq_collector_new_buffer(self.handle, C.byref(h)))
        return Collector(self, h.value, "buffer")

    def grid_collector(self, bmin, bmax, cell: float) -> Collector:
        h = C.c_void_p()
        _check(self.lib.pcq_collector_new_grid(self.handle, _d3(bmin), _d3(bmax), float(cell), C.byref(h)))
        return Collector(self, h.value, "grid")

    # memory -----------------------------------------------------------------------------------
    def alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        _check(self.lib.pcq_device_alloc(self.handle, int(nbytes), C.byref(p)))
        return p.value

    def free(self, ptr: int) -> None:
        _check(self.lib.pcq_device_free(self.handle, C.c_void_p(ptr)))

    def to_device(self, dst: int, src: np.ndarray) -> None:
        src = np.ascontiguousarray(src)
        _check(self.lib.pcq_copy_to_device(self.handle, C.c_void_p(dst), src.ctypes.data_as(C.c_void_p), src.nbytes))

    def to_host(self, dst: np.ndarray, src: int) -> None:
        _check(self.lib.pcq_copy_to_host(self.handle, dst.ctypes.data_as(C.c_void_p), C.c_void_p(src), dst.nbytes))

    def memset(self, dst: int, value: int, nbytes: int, stream: Optional[int] = None) -> None:
        _check(self.lib.pcq_device_memset(self.handle, C.c_void_p(dst), value, nbytes, C.c_void_p(stream)))

    # scans ------------------------------------------------------------------------------------
    def scan_dev(self, cols: Columns, pred: Predicate, coll: Collector, stream: Optional[int] = None) -> None:
        _check(self.lib.pcq_scan_dev(self.handle, C.byref(cols), C.byref(pred), coll.handle, C.c_void_p(stream)))

    def scan_host(self, cols: Columns, pred: Predicate, coll: Collector) -> None:
        _check(self.lib.pcq_scan_host(self.handle, C.byref(cols), C.byref(pred), coll.handle))

    def scan_host_nowait(self, cols: Columns, pred: Predicate, coll: Collector) -> None:
        """scan_host that returns once the caller's columns have been read; results after synchronize() / accessors."""
        _check(self.lib.pcq_scan_host_nowait(self.handle, C.byref(cols), C.byref(pred), coll.handle))

    def scan_fd(self, fd: int, cols: Columns, pred: Predicate, coll: Collector) -> None:
        """Like scan_host, with the column pointers given as byte offsets into the open file `fd`."""
        _check(self.lib.pcq_scan_fd(self.handle, fd, C.byref(cols), C.byref(pred), coll.handle))

    def scan_fd_nowait(self, fd: int, cols: Columns, pred: Predicate, coll: Collector) -> None:
        """scan_fd that returns once the file has been read and its last kernels are enqueued; results after synchronize() / accessors."""
        _check(self.lib.pcq_scan_fd_nowait(self.handle, fd, C.byref(cols), C.byref(pred), coll.handle))

    def prepare_host_scans(self) -> None:
        """Starts pinning the staging ring and the copy helpers on a thread of the library (returns at once)."""
        _check(self.lib.pcq_prepare_host_scans(self.handle))

    def scan_dev_count_batch(self, cols: Sequence[Columns], preds: Sequence[Predicate], device_total: int,
                             stream: Optional[int] = None) -> None:
        n = len(cols)
        ca = (Columns * n)(*cols)
        pa = (Predicate * n)(*preds)
        _check(self.lib.pcq_scan_dev_count_batch(self.handle, ca, pa, n, C.c_void_p(device_total), C.c_void_p(stream)))

    # chunk index --------------------------------------------------------------------------------
    def index_new(self) -> int:
        h = C.c_void_p()
        _check(self.lib.pcq_index_new(self.handle, C.byref(h)))
        return h.value

    def index_free(self, ix: int) -> None:
        self.lib.pcq_index_free(C.c_void_p(ix))

    def index_stats(self, ix: int) -> dict:
        st = IndexStats()
        _check(self.lib.pcq_index_get_stats(C.c_void_p(ix), C.byref(st)))
        return {k: getattr(st, k) for k, _ in IndexStats._fields_}

    def scan_dev_indexed(self, cols: Columns, pred: Predicate, ix: int, coll: Collector, stream: Optional[int] = None) -> None:
        _check(self.lib.pcq_scan_dev_indexed(self.handle, C.byref(cols), C.byref(pred), C.c_void_p(ix), coll.handle,
                                             C.c_void_p(stream)))

    def synth_fill(self, spec: SynthSpec, first: int, count: int, d_xyz: Optional[int], d_cls: Optional[int],
                   stream: Optional[int] = None) -> None:
        _check(self.lib.pcq_synth_fill_dev(self.handle, C.byref(spec), first, count, C.c_void_p(d_xyz),
                                           C.c_void_p(d_cls), C.c_void_p(stream)))


def make_columns(xyz=None, cls=None, rgb=None, n=0, xyz_stride=12, cls_stride=1, rgb_stride=6, first_index=0,
                 scale=(1.0, 1.0, 1.0), offset=(0.0, 0.0, 0.0)) -> Columns:
    c = Columns()
    c.xyz, c.cls, c.rgb = xyz, cls, rgb
    c.xyz_stride, c.cls_stride, c.rgb_stride = xyz_stride, cls_stride, rgb_stride
    c.n, c.first_index = n, first_index
    for a in range(3):
        c.scale[a] = float(scale[a])
        c.offset[a] = float(offset[a])
    return c
