import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "dgcnn_golden.npz"))


@pytest.fixture(scope="session")
def att_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "attention_golden.npz"))


@pytest.fixture(scope="session")
def dev():
    """cuda:0; GPU tests FAIL (not skip) when the HIP library is missing on a GPU box."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU in this container")
    from gcanet_amd import _lib, build
    # mtime-incremental (no-op when current): a .hip edit can never be tested against a stale library.  A build step,
    # not a fallback -- there is no other implementation to fall back to.
    if os.path.exists(build.HIPCC):
        build.build(verbose=False)
    elif build.stale():
        pytest.fail("libgcanet_hip.so is older than its sources and hipcc is not available to rebuild it")
    _lib.lib()  # raises if libgcanet_hip.so is absent -> loud failure, no silent fallback
    return torch.device("cuda:0")
