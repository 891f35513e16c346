import importlib
import os
import sys

import numpy as np
import pytest

try:  # torch first: its bundled HIP runtime must be the one in the process.  Collecting tests/test_gpu_dropin.py dlopens
    import torch  # noqa: F401  (oracle/_ref/*.so -> liblvba_hip.so -> the SYSTEM libamdhip64; if that loads before torch's own copy,
except Exception:  # a later torch.cuda call and the library disagree about the device (seen as "no HIP device available"))
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("global-lvba_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("global-lvba_amd.synth")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build_c()
    return oracle


_PROBLEMS = {}


def make_problem(n_poses, n_voxels, **kw):
    """Cached synthetic problems (CPU generation)."""
    key = (n_poses, n_voxels, tuple(sorted(kw.items())))
    if key not in _PROBLEMS:
        synth = importlib.import_module("global-lvba_amd.synth")
        _PROBLEMS[key] = synth.make_balm_problem(n_poses, n_voxels, **kw)
    return _PROBLEMS[key]


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
