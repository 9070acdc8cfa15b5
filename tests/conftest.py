"""Test configuration.

* ``-m "not gpu"``: oracle vs golden vectors, host logic, ABI symbol checks — no GPU call.
* ``-m gpu``: parity of the HIP path (through the C ABI) against the oracle, on an MI355X.
"""
import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


def _build_oracle():
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("pcq_oracle.c", "lazer_oracle.c", "synth.c", "pcq_oracle.h", "query_oracle.c")]
    exe = os.path.join(ROOT, "oracle", "query_oracle")
    newest = max(os.path.getmtime(s) for s in srcs)
    if not (os.path.exists(so) and os.path.exists(exe)) or min(os.path.getmtime(so), os.path.getmtime(exe)) < newest:
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], check=True, capture_output=True)


@pytest.fixture(scope="session")
def oracle():
    _build_oracle()
    import _oracle
    return _oracle.Oracle()


@pytest.fixture(scope="session")
def pcq():
    """The product's ctypes view (package directory has hyphens -> importlib)."""
    return importlib.import_module("adhoc-queries-pointclouds_amd")


@pytest.fixture(scope="session")
def gpu_ctx(pcq):
    ctx = pcq.Context(0)
    yield ctx
    ctx.close()
