import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
# The staged SpMV streams dictionary-coded columns only for operators of >= 4 M entries by default (below that an
# iteration is latency bound); the tests' operators are small, so every context of the test session (and of the processes
# it spawns) forces the coded stream wherever an operator qualifies.  tests/test_gpu_primitives.py checks the default too.
os.environ.setdefault("KHIP_SPMV_CODES", "2")
# ... and likewise the block-delta column stream of the stream kernel (csrc/coldelta.hip): whatever the operator's size
os.environ.setdefault("KHIP_SPMV_DELTA", "2")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as ok  # oracle/oracle.py -- the checker, never the product
    ok.lib()
    return ok


@pytest.fixture(scope="session")
def K():
    import krylov_jl_amd as K
    K.lib()
    return K


@pytest.fixture(scope="session")
def ctx(K):
    if not K.gpu_available():
        pytest.fail("GPU test selected but /dev/kfd is absent: the HIP path has no CPU fallback")
    c = K.Context(0)
    yield c
    # no lazily built accelerator may have failed for a reason other than a full device (such a failure falls back
    # to the plain kernels, so result-only tests would still pass)
    import ctypes
    cnt = ctypes.c_int(-1)
    assert K.lib().khip_test_optional_build_failures(ctypes.byref(cnt)) == 0 and cnt.value == 0, cnt.value
    c.close()


_PARITY_LOG = os.path.join(ROOT, "gpurun_out", "parity_log.jsonl")


@pytest.fixture(scope="session")
def parity_log():
    """Append measured GPU-vs-oracle deviations to gpurun_out/parity_log.jsonl (evidence for DESIGN.md)."""
    os.makedirs(os.path.dirname(_PARITY_LOG), exist_ok=True)

    def log(**kw):
        with open(_PARITY_LOG, "a") as f:
            f.write(json.dumps(kw) + "\n")
    return log
