import os
import sys

import pytest

# the HIP runtime's hardware queues are the HOST's to choose (the library no longer sets the variable behind its host's back: include/rptr_hip.h
# RPTR_CREATE_SET_HW_QUEUES); this host -- the test process -- wants one per frame context, before its first HIP call
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "library_defaults: the test runs on the library's own defaults (no RPTR_* override of any option)")


@pytest.fixture(autouse=True)
def _instances_stay_two_level_unless_a_test_says_otherwise(request, monkeypatch):
    """The library's default builds a static multi-instance scene as ONE world-space tree (option "flatten" = auto, include/rptr_hip.h
    "Options"): hits are then found on pre-transformed triangles and t / u / v differ from the object-space walk of the reference's
    TLAS / BLAS by rounding. The parity tests pin the two-level form -- bit for bit against the oracle's object-space brute force -- so the
    suite runs with the option's environment override RPTR_FLATTEN=0 unless a test sets the variable itself (the flattened form: oracle on
    the exported tree) or is marked `library_defaults` (the whole-frame C3 / C4 tests, the option tests): those run on what a host that
    sets nothing gets."""
    if "library_defaults" in request.keywords:
        monkeypatch.delenv("RPTR_FLATTEN", raising=False)
    elif "RPTR_FLATTEN" not in os.environ:
        monkeypatch.setenv("RPTR_FLATTEN", "0")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib


@pytest.fixture(scope="session")
def hip_lib():
    """The HIP C-ABI library; GPU tests fail loudly when it is missing (no fallback)."""
    from realtimepathtracingresearchframework_amd import backend
    return backend.load_library()
