"""The iterations inside ceres::Solve (reference src/lvba_system.cpp:1643) against the REAL Ceres Solver -- the one piece of the
visual path this repository cannot pin by itself (no Ceres / Eigen in the build image or on the GPU box: DESIGN.md section 2).
Skipped unless tools/pin_ceres/pin_ceres.sh has built its driver against an installed Ceres (LVBA_CERES_PIN_DRIVER or
tools/pin_ceres/ceres_pin_driver): then lvba_visual_refine's per-iteration trace and refined cameras are held against the real
solver's on the same problem, to north_star's 1e-5."""
import importlib
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.environ.get("LVBA_CERES_PIN_DRIVER", os.path.join(ROOT, "tools", "pin_ceres", "ceres_pin_driver"))


def write_problem(path, d):
    """the directory layout tools/pin_ceres/ceres_pin_driver.cpp reads"""
    os.makedirs(path, exist_ok=True)
    M, T, O = len(d["q"]), len(d["X"]), int(d["obs_off"][-1])
    with open(os.path.join(path, "meta.txt"), "w") as f:
        f.write(f"{M} {T} {O}\n")
    for name, arr, dt in (("q", d["q"], np.float64), ("t", d["t"], np.float64), ("X", d["X"], np.float64),
                          ("obs_off", d["obs_off"], np.int64), ("obs_cam", d["obs_cam"], np.int32), ("obs_uv", d["obs_uv"], np.float64),
                          ("plane", d["plane"], np.float64), ("valid", d["valid"], np.int32), ("intr", d["intr"], np.float64)):
        np.ascontiguousarray(arr, dtype=dt).tofile(os.path.join(path, name + ".bin"))
    return M, T, O


def test_problem_directory_layout(tmp_path):
    """(CPU) what the driver reads is what write_problem writes: sizes and dtypes of every file"""
    rng = np.random.default_rng(0)
    M, T = 3, 5
    off = np.array([0, 2, 4, 7, 9, 12], np.int64)
    d = dict(q=rng.standard_normal((M, 4)), t=rng.standard_normal((M, 3)), X=rng.standard_normal((T, 3)), obs_off=off,
             obs_cam=rng.integers(0, M, 12), obs_uv=rng.standard_normal((12, 2)), plane=rng.standard_normal((T, 4)),
             valid=np.ones(T, np.int32), intr=np.arange(8.0))
    assert write_problem(str(tmp_path / "p"), d) == (M, T, 12)
    sizes = {"q": 32 * M, "t": 24 * M, "X": 24 * T, "obs_off": 8 * (T + 1), "obs_cam": 48, "obs_uv": 192, "plane": 32 * T, "valid": 4 * T, "intr": 64}
    for k, v in sizes.items():
        assert os.path.getsize(tmp_path / "p" / (k + ".bin")) == v, k
    assert np.array_equal(np.fromfile(tmp_path / "p" / "obs_off.bin", np.int64), off)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(DRIVER), reason="no Ceres-linked driver: run tools/pin_ceres/pin_ceres.sh on a machine with Ceres 2.1")
def test_visual_refine_matches_real_ceres(tmp_path):
    pkg = importlib.import_module("global-lvba_amd")
    synth = importlib.import_module("global-lvba_amd.synth")
    d = synth.make_visual_problem(40, 1500, rot_sigma_deg=0.3, trans_sigma=0.10, point_sigma=0.30, device="cuda")
    write_problem(str(tmp_path / "p"), d)
    r = subprocess.run([DRIVER, str(tmp_path / "p")], capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = json.loads(r.stdout.strip().splitlines()[-1])
    prob = pkg.VisualProblem(len(d["q"]), d["obs_off"], d["obs_cam"], d["obs_uv"], d["plane"], d["valid"], d["intr"])
    (q, t, X), trace, term, rc = prob.refine(d["q"], d["t"], d["X"])
    prob.close()
    assert rc == 0
    its = ref["iterations"]
    assert len(trace) == len(its), (len(trace), len(its), ref["ceres_version"])
    for row, it in zip(trace, its):
        assert row["accepted"] == it["accepted"], (row, it)
        assert abs(row["cost"] - it["cost"]) <= 1e-5 * abs(it["cost"]), (row, it)
        assert abs(row["radius"] - it["radius"]) <= 1e-5 * abs(it["radius"]), (row, it)
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    qr = np.array(ref["q"])
    qr /= np.linalg.norm(qr, axis=1, keepdims=True)
    assert np.abs(qn - qr).max() <= 1e-5 and np.abs(t - np.array(ref["t"])).max() <= 1e-5
