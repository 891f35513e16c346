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


def ceres_probe():
    """Is there a Ceres Solver (the one dependency whose iterations the visual path cannot pin by itself) on this
    machine?  Cheap: the usual include / library directories and the loader cache, no walk of the file system."""
    import glob
    import subprocess
    hits = []
    for pat in ("/usr/include/ceres/ceres.h", "/usr/local/include/ceres/ceres.h", "/opt/*/include/ceres/ceres.h",
                "/usr/lib/*/libceres*", "/usr/lib/libceres*", "/usr/local/lib/libceres*", "/opt/*/lib/libceres*",
                "/usr/lib/cmake/Ceres*", "/usr/lib/*/cmake/Ceres*", "/usr/local/lib/cmake/Ceres*"):
        hits += glob.glob(pat)
    try:
        out = subprocess.run(["ldconfig", "-p"], capture_output=True, text=True, timeout=10).stdout
        hits += [ln.strip() for ln in out.splitlines() if "libceres" in ln]
    except Exception:
        pass
    return hits


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """One line per run (also under -q): why tests/test_gpu_ceres_pin.py::test_visual_refine_matches_real_ceres ran or skipped."""
    if "gpu" not in (config.getoption("-m") or "") or "not gpu" in (config.getoption("-m") or ""):
        return
    hits = ceres_probe()
    driver = os.environ.get("LVBA_CERES_PIN_DRIVER", os.path.join(ROOT, "tools", "pin_ceres", "ceres_pin_driver"))
    terminalreporter.write_line(
        "ceres probe: " + (f"found {hits[:3]}" if hits else "no Ceres Solver on this machine (headers, libraries, loader cache)")
        + f"; pin driver {'present' if os.path.exists(driver) else 'absent'} -> test_visual_refine_matches_real_ceres "
        + ("runs" if os.path.exists(driver) else "SKIPPED: the iterations inside ceres::Solve stay unpinned (tools/pin_ceres/pin_ceres.sh builds the driver where Ceres 2.1 exists)"))


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


class HostTransport:
    """tests/host_transport.cpp: ranks = host threads of this process on one GPU, all-reduces staged through the host and
    reduced in rank order.  TEST INFRASTRUCTURE (the product only sees an lvba_allreduce_fn callback).  Built by
    __graft_entry__.build() into tests/_build/libhost_transport.so."""
    _lib = None

    @classmethod
    def lib(cls):
        import ctypes as C
        if cls._lib is None:
            path = os.path.join(ROOT, "tests", "_build", "libhost_transport.so")
            if not os.path.exists(path):
                import __graft_entry__ as g
                g.build_test_transport()
            lib = C.CDLL(path)
            lib.ht_create.restype = C.c_void_p
            lib.ht_create.argtypes = [C.c_int]
            lib.ht_rank.restype = C.c_void_p
            lib.ht_rank.argtypes = [C.c_void_p, C.c_int]
            lib.ht_rank_free.argtypes = [C.c_void_p]
            lib.ht_destroy.argtypes = [C.c_void_p]
            lib.ht_poison.argtypes = [C.c_void_p]
            lib.ht_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
            cls._lib = lib
        return cls._lib

    def __init__(self, world):
        import ctypes as C
        self.world = world
        self.comm = self.lib().ht_create(world)
        self.fn = C.cast(self.lib().ht_allreduce, C.c_void_p).value
        self.ctx = [self.lib().ht_rank(self.comm, r) for r in range(world)]

    def attach(self, prob, rank):
        prob.dist_init_external(self.world, rank, self.fn, self.ctx[rank])

    def poison(self):
        self.lib().ht_poison(self.comm)

    def stats(self):
        import ctypes as C
        c, b = C.c_uint64(), C.c_uint64()
        self.lib().ht_stats(self.comm, C.byref(c), C.byref(b))
        return int(c.value), int(b.value)

    def run(self, rank_main, timeout=600):
        """rank_main(r) in one thread per rank; a rank that raises poisons the communicator so the others return instead of
        hanging in a barrier.  Returns the list of results; re-raises the first error."""
        import threading
        out, err = [None] * self.world, [None] * self.world

        def wrap(r):
            try:
                out[r] = rank_main(r)
            except BaseException as e:
                err[r] = e
                self.poison()

        th = [threading.Thread(target=wrap, args=(r,), daemon=True) for r in range(self.world)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=timeout)
        first = next((e for e in err if e is not None and "external all-reduce failed" not in str(e)), None) or \
            next((e for e in err if e is not None), None)
        if first is not None:
            raise first
        assert all(o is not None for o in out), "a rank did not finish (collective mismatch?)"
        return out
