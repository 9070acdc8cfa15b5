"""adhoc-queries-pointclouds_amd — MI355X-native predicate path for ad-hoc point-cloud queries.

The product is native code:
  * ``libpcq.so``        HIP kernels + the thin C ABI of ``include/pcq.h`` (``csrc/``)
  * ``libpcq_query.so``  C++ host layer mirroring the reference's Searcher / ResultCollector
                         interface (``host/``), plus the ``query`` CLI binary

This Python package is only a ctypes view of those C ABIs, used by the tests, ``bench.py`` and
``__graft_entry__.py``.  It never computes a scan itself and has no CPU fallback: if the HIP
library is missing or no device is usable, it raises.

The directory name contains hyphens (it mirrors the upstream repository name), so import it with
``importlib.import_module("adhoc-queries-pointclouds_amd")``.
"""
from .binding import (  # noqa: F401
    PcqError,
    Columns,
    Predicate,
    Point,
    SynthSpec,
    POINT_DTYPE,
    Context,
    Collector,
    lib_path,
    load_library,
    hip_runtime_path,
    box_to_local,
    exported_symbols,
    declared_symbols,
)
