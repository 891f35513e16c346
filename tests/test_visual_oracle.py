"""CPU tests (no GPU) of the visual-stage oracle and of the hand-derived device math.

PARITY UNPINNED against Ceres itself (not installed, not in /root/reference): the oracle is pinned by
self-consistency -- autograd Jacobians vs finite differences through the manifold, Schur elimination vs the full normal
equations, Plus/PlusJacobian consistency, the reference's quirks (camera 0 constant, plane-less landmarks dropped,
z <= 1e-8 -> zero residual) -- and frozen by tests/golden/visual_small.npz."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT, rel
from oracle import visual_oracle as vo


def _problem(synth, **kw):
    d = synth.make_visual_problem(**kw)
    p = vo.VisualProblem(d["q"], d["t"], d["X"], d["obs_off"], d["obs_cam"], d["obs_uv"], d["plane"], d["valid"], d["intr"])
    return d, vo.VisualOracle(p)


def test_jacobian_matches_finite_differences_through_the_manifold(synth):
    d, o = _problem(synth, n_cams=6, n_tracks=30, seed=7)
    q, t, X = o.state()
    r, J = o.residuals_and_jacobian(q, t, X)
    rng = np.random.default_rng(0)
    for _ in range(3):
        dl = rng.standard_normal(o.n_par) * 1e-6
        rp, _ = o.residuals_and_jacobian(*o.plus(q, t, X, dl), want_jac=False)
        rm, _ = o.residuals_and_jacobian(*o.plus(q, t, X, -dl), want_jac=False)
        assert np.abs((rp - rm) / 2 - J @ dl).max() <= 1e-7 * np.abs(J @ dl).max()


def test_manifold_plus_and_jacobian_are_consistent():
    rng = np.random.default_rng(1)
    for _ in range(10):
        a = rng.standard_normal(4); a /= np.linalg.norm(a)
        dlt = rng.standard_normal(3) * 0.3
        out = vo.eigen_quat_plus(a, dlt)
        assert abs(np.linalg.norm(out) - 1) <= 1e-15                       # stays on the unit sphere
        h = 1e-7
        fd = np.stack([(vo.eigen_quat_plus(a, h * e) - vo.eigen_quat_plus(a, -h * e)) / (2 * h) for e in np.eye(3)], 1)
        assert np.abs(fd - vo.eigen_quat_plus_jacobian(a)).max() <= 1e-9
    assert np.array_equal(vo.eigen_quat_plus(a, np.zeros(3)), a)


def test_schur_elimination_equals_full_normal_equations(synth):
    d, o = _problem(synth, n_cams=6, n_tracks=30, seed=8)
    r, J = o.residuals_and_jacobian(*o.state())
    D = np.sqrt(np.clip((J * J).sum(0), 1e-6, 1e32) / 50.0)
    x = o.solve_schur(J, r, D)
    x_full = np.linalg.solve(J.T @ J + np.diag(D * D), J.T @ r)
    assert rel(x, x_full) <= 1e-9


def test_reference_quirks(synth):
    d, o = _problem(synth, n_cams=6, n_tracks=40, seed=5, invalid_frac=0.3)
    assert len(o.act) == int(d["valid"].sum()) < 40                          # plane-less landmarks are dropped ...
    n_obs_kept = sum(int(d["obs_off"][i + 1] - d["obs_off"][i]) for i in o.act)
    assert len(o.rows) == n_obs_kept + len(o.act)                            # ... with their observations; 1 plane row each
    assert o.n_cam == 6 * 5                                                  # camera 0 constant
    r, J = o.residuals_and_jacobian(*o.state())
    pos, plane_idx = 0, []
    for row in o.rows:                                                       # reprojection rows hold 2 residuals, plane rows 1
        if row[0] == "p":
            plane_idx.append(pos)
        pos += 2 if row[0] == "r" else 1
    assert pos == len(r) and np.all(r[plane_idx] >= 0)                        # sqrt(s^2+1e-12)/sigma
    # a landmark behind its camera: zero residual (utils.hpp:78)
    F64 = torch.float64
    rr = vo.reproj_residual(torch.tensor([1.0, 0, 0, 0], dtype=F64), torch.zeros(3, dtype=F64), torch.tensor([0.3, 0.2, -1.0], dtype=F64),
                            torch.zeros(2, dtype=F64), [float(v) for v in d["intr"]], 0.5)
    assert float(rr.abs().max()) == 0.0


def test_lm_converges_and_reduces_the_error(synth):
    d, o = _problem(synth, n_cams=8, n_tracks=60, seed=3)
    (q, t, X), trace, status = o.solve()
    assert status.startswith("CONVERGENCE")
    assert trace[-1]["cost"] < 0.02 * trace[0]["cost"]
    assert np.abs(t - d["t_gt"]).max() < np.abs(d["t"] - d["t_gt"]).max()
    assert np.array_equal(t[0], d["t"][0]) and np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-15)


def test_oracle_reproduces_golden(synth):
    z = np.load(os.path.join(ROOT, "tests", "golden", "visual_small.npz"))
    p = vo.VisualProblem(z["q"], z["t"], z["X"], z["obs_off"], z["obs_cam"], z["obs_uv"], z["plane"], z["valid"], z["intr"])
    o = vo.VisualOracle(p)
    assert abs(o.cost(*o.state()) - z["cost0"]) <= 1e-12 * z["cost0"]
    (q, t, X), trace, status = o.solve()
    assert len(trace) == len(z["trace_cost"]) and rel([r["cost"] for r in trace], z["trace_cost"]) <= 1e-9
    assert np.abs(t - z["t_final"]).max() <= 1e-9 and np.abs(X - z["X_final"]).max() <= 1e-9


# ---------------------------------------------------------------------------------------------- device math on the host
@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("emulv") / "libemul.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", os.path.join(ROOT, "tests", "host_emul.cpp"), "-o", so])
    lib = ctypes.CDLL(so)
    f64p = np.ctypeslib.ndpointer(np.float64, flags="C")
    lib.emul_reproj.argtypes = [f64p] * 5 + [ctypes.c_double] + [f64p] * 3
    lib.emul_plane.argtypes = [f64p, f64p, ctypes.c_double, f64p]
    lib.emul_plane.restype = ctypes.c_double
    lib.emul_quat_plus.argtypes = [f64p] * 3
    return lib


def test_device_reprojection_jacobians_match_autograd(emul, synth):
    """The kernels' hand-derived Jacobians (visual_math.h) == autograd of the reference functor chained with the
    (mis-ordered) EigenQuaternionManifold PlusJacobian."""
    d = synth.make_visual_problem(8, 60, seed=3)
    intr = d["intr"]
    F64 = torch.float64
    for o in range(0, len(d["obs_cam"]), 5):
        c = int(d["obs_cam"][o]); ti = int(np.searchsorted(d["obs_off"], o, side="right") - 1)
        q, t, X, uv = d["q"][c].copy(), d["t"][c].copy(), d["X"][ti].copy(), d["obs_uv"][o].copy()
        r, Jc, Jp = np.zeros(2), np.zeros(12), np.zeros(6)
        assert emul.emul_reproj(q, t, X, uv, intr, 0.5, r, Jc, Jp) == 1
        qt, tt, Xt = (torch.tensor(v, dtype=F64, requires_grad=True) for v in (q, t, X))
        rr = vo.reproj_residual(qt, tt, Xt, torch.tensor(uv, dtype=F64), [float(v) for v in intr], 0.5)
        Jrc, Jrp = np.zeros((2, 6)), np.zeros((2, 3))
        for k in range(2):
            gq, gt, gX = torch.autograd.grad(rr[k], (qt, tt, Xt), retain_graph=True)
            Jrc[k, :3] = gq.numpy() @ vo.eigen_quat_plus_jacobian(q); Jrc[k, 3:] = gt.numpy(); Jrp[k] = gX.numpy()
        assert np.abs(r - rr.detach().numpy()).max() <= 1e-10 * max(1.0, np.abs(r).max())
        assert rel(Jc.reshape(2, 6), Jrc) <= 1e-11 and rel(Jp.reshape(2, 3), Jrp) <= 1e-11
    # behind the camera
    r, Jc, Jp = np.ones(2), np.ones(12), np.ones(6)
    assert emul.emul_reproj(np.array([1.0, 0, 0, 0]), np.zeros(3), np.array([0.0, 0.0, -1.0]), np.zeros(2), intr, 0.5, r, Jc, Jp) == 0
    assert not r.any() and not Jc.any() and not Jp.any()
    # plane prior + manifold plus
    J = np.zeros(3)
    X, pl = d["X"][0].copy(), d["plane"][0].copy()
    rp = emul.emul_plane(X, pl, 0.01, J)
    Xt = torch.tensor(X, dtype=F64, requires_grad=True)
    rr = vo.plane_residual(Xt, torch.tensor(pl, dtype=F64), 0.01)
    (g,) = torch.autograd.grad(rr, (Xt,))
    assert abs(rp - float(rr)) <= 1e-12 * rp and np.abs(J - g.numpy()).max() <= 1e-12 * np.abs(J).max()
    rng = np.random.default_rng(0)
    for _ in range(5):
        a = rng.standard_normal(4); a /= np.linalg.norm(a); dd = rng.standard_normal(3) * 0.1; out = np.zeros(4)
        emul.emul_quat_plus(a, dd, out)
        assert np.abs(out - vo.eigen_quat_plus(a, dd)).max() <= 1e-15


def test_converged_answer_does_not_depend_on_the_solver(synth):
    """The iterations of ceres::Solve are restated from memory (unpinned).  What they converge TO is not a matter of the
    solver: an independent method -- undamped Gauss-Newton on the full normal equations (numpy lstsq) with step halving, no
    Jacobi scaling, no trust region, no Schur complement -- reaches the same minimum of the same pinned cost, and the
    Ceres-style LM of the oracle stops within its function tolerance (1e-6 relative) of it."""
    d, o = _problem(synth, n_cams=6, n_tracks=40, seed=5)
    (q, t, X), trace, status = o.solve()
    assert status.startswith("CONVERGENCE")
    q2, t2, X2 = o.state()
    cost = o.cost(q2, t2, X2)
    for _ in range(40):
        r, J = o.residuals_and_jacobian(q2, t2, X2)
        dx = np.linalg.lstsq(J, -r, rcond=None)[0]
        step = 1.0
        while True:
            cand = o.plus(q2, t2, X2, step * dx)
            c = o.cost(*cand)
            if c < cost or step < 1e-6:
                break
            step *= 0.5
        if c >= cost:
            break
        done = (cost - c) < 1e-14 * cost
        (q2, t2, X2), cost = cand, c
        if done:
            break
    lm_cost = trace[-1]["cost"]
    assert cost <= lm_cost * (1 + 1e-12) and (lm_cost - cost) < 1e-5 * cost
    act = o.act
    # the LM stops on its function tolerance, a little before the minimum: poses agree to what that leaves
    assert np.abs(t - t2).max() < 2e-3 * max(1.0, np.abs(t2).max()) and np.abs(X[act] - X2[act]).max() < 5e-3
    assert np.abs(np.abs(np.einsum("ij,ij->i", q, q2)) - 1).max() < 1e-6
