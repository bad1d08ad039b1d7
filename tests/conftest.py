import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        import torch
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiu" else z[k]) for k in z.files}
    return load


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(autouse=True)
def _scribble_lds_between_launches(request, monkeypatch):
    """SNERF_TEST_SCRIBBLE_LDS=1 python -m pytest tests -m gpu: every library call of every GPU test is preceded by snerf_debug_lds_scribble
    (a new seed each time), so that a kernel reading LDS it has not written compares against its oracle / golden with garbage instead of its
    previous launch's leftovers (tests/test_stale_lds.py is the always-on form for whole passes).  Off by default: it costs a launch per call."""
    if os.environ.get("SNERF_TEST_SCRIBBLE_LDS", "") == "" or request.node.get_closest_marker("gpu") is None or not has_gpu():
        yield
        return
    from snerf_amd import _lib, ops
    real = _lib.call
    n = [0]

    def call(name, *args):
        if name != "snerf_debug_lds_scribble" and ops.__dict__.get("_stream") is not None:
            n[0] += 1
            real("snerf_debug_lds_scribble", (n[0] * 2654435761) & 0x7fffffff, ops._stream())
        return real(name, *args)
    monkeypatch.setattr(_lib, "call", call)
    yield
