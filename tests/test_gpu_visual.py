"""GPU parity tests of the visual stage (lvba_visual_*) against oracle/visual_oracle.py on identical inputs.

The oracle differentiates the reference's functors with torch autograd (what Ceres' Jets do) and runs a restatement of
Ceres 2.1's LM schedule; the GPU path uses hand-derived Jacobians, the Schur complement in its rank-3 Y form, the
shared pair-assembly pass and the LDL^T solver.  fp64 throughout; tolerances 1e-9 on costs/systems, 1e-7 on traces."""
import numpy as np
import pytest

from conftest import rel

pytestmark = pytest.mark.gpu

CASES = [dict(n_cams=8, n_tracks=60, seed=3), dict(n_cams=20, n_tracks=300, seed=4, track_len=5),
         dict(n_cams=6, n_tracks=40, seed=5, invalid_frac=0.3)]


def _mk(pkg, synth, case):
    from oracle import visual_oracle as vo
    d = synth.make_visual_problem(**case)
    prob = pkg.VisualProblem(d["q"].shape[0], d["obs_off"], d["obs_cam"], d["obs_uv"], d["plane"], d["valid"], d["intr"])
    orc = vo.VisualOracle(vo.VisualProblem(d["q"], d["t"], d["X"], d["obs_off"], d["obs_cam"], d["obs_uv"], d["plane"],
                                           d["valid"], d["intr"]))
    return d, prob, orc


@pytest.mark.parametrize("case", CASES)
def test_cost_matches_oracle(pkg, synth, case):
    d, prob, orc = _mk(pkg, synth, case)
    for q, t, X in ((d["q"], d["t"], d["X"]), (d["q_gt"], d["t_gt"], d["X_gt"])):
        c_ref = orc.cost(q, t, X)
        assert abs(prob.cost(q, t, X) - c_ref) <= 1e-10 * c_ref


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("radius", [1e4, 3.0])
def test_reduced_camera_system_matches_oracle(pkg, synth, case, radius):
    """S = B + D^2 - E (C + D^2)^-1 E^T and its right-hand side in the Jacobi-scaled tangent variables."""
    d, prob, orc = _mk(pkg, synth, case)
    q, t, X = orc.state()
    r, J = orc.residuals_and_jacobian(q, t, X)
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0)))
    J = J * scale
    D2 = np.clip((J * J).sum(0), 1e-6, 1e32) / radius
    A = J.T @ J + np.diag(D2)
    g = J.T @ r
    nc = orc.n_cam
    B, E, C = A[:nc, :nc], A[:nc, nc:], A[nc:, nc:]
    Ci = np.linalg.inv(C)          # block diagonal
    S_ref = B - E @ Ci @ E.T
    rhs_ref = g[:nc] - E @ (Ci @ g[nc:])
    S, rhs, c = prob.linearize(q, t, X, radius)
    assert abs(c - 0.5 * r @ r) <= 1e-10 * (0.5 * r @ r)
    assert rel(S[6:, 6:], S_ref) <= 1e-9
    assert rel(rhs[6:], rhs_ref) <= 1e-9
    # camera 0 is constant: decoupled block, zero right-hand side
    assert np.abs(S[:6, 6:]).max() == 0.0 and np.abs(rhs[:6]).max() == 0.0
    assert np.array_equal(S, S.T)


@pytest.mark.parametrize("case", CASES)
def test_refine_matches_oracle(pkg, synth, case):
    d, prob, orc = _mk(pkg, synth, case)
    (q, t, X), trace, term, rc = prob.refine(d["q"], d["t"], d["X"])
    (qr, tr, Xr), trace_ref, term_ref = orc.solve()
    assert rc == 0
    assert term == term_ref
    assert len(trace) == len(trace_ref)
    for a, b in zip(trace, trace_ref):
        assert a["accepted"] == b["accepted"]
        assert abs(a["cost"] - b["cost"]) <= 1e-7 * abs(b["cost"])
        assert abs(a["radius"] - b["radius"]) <= 1e-6 * b["radius"]
    assert np.abs(q - qr).max() <= 1e-8 and np.abs(t - tr).max() <= 1e-7 and np.abs(X - Xr).max() <= 1e-7
    # landmarks without a plane are returned untouched (src/lvba_system.cpp:1598-1603)
    inv = d["valid"] == 0
    assert np.array_equal(X[inv], d["X"][inv])
    # camera 0 is constant
    assert np.array_equal(t[0], d["t"][0])


def test_refine_with_rejected_steps_matches_oracle(pkg, synth):
    """A start far enough from the optimum that trust-region steps are REJECTED in the middle of the run (iterations 4-7 of
    12): after a rejection the next system is linearised at the same point with a smaller radius, so the residuals of the
    rejected trial point must not have replaced the ones of the linearisation point (they did until round 2: every later step
    was garbage and the run ended on the radius test).  Row by row against the oracle."""
    case = dict(n_cams=8, n_tracks=60, seed=3, rot_sigma_deg=1.5, trans_sigma=0.3, point_sigma=0.6)
    d, prob, orc = _mk(pkg, synth, case)
    (q, t, X), trace, term, rc = prob.refine(d["q"], d["t"], d["X"])
    (qr, tr, Xr), trace_ref, term_ref = orc.solve()
    acc = [b["accepted"] for b in trace_ref]
    assert 0 in acc[1:-1] and acc[-2:] != [0, 0]                     # rejections followed by accepted steps
    assert rc == 0 and term == term_ref and len(trace) == len(trace_ref)
    for a, b in zip(trace, trace_ref):
        assert a["accepted"] == b["accepted"]
        assert abs(a["cost"] - b["cost"]) <= 1e-6 * abs(b["cost"])
        assert abs(a["radius"] - b["radius"]) <= 1e-6 * b["radius"]
    assert np.abs(q - qr).max() <= 1e-7 and np.abs(t - tr).max() <= 1e-6 and np.abs(X - Xr).max() <= 1e-6


def test_reference_entry_mirror_and_edges(pkg, synth):
    d = synth.make_visual_problem(8, 60, seed=3)
    n = d["plane"][:, :3].copy()
    (q, t, X), trace, term, rc, valid = pkg.optimize_camera_poses(d["q"], d["t"], d["X"], d["obs_off"], d["obs_cam"], d["obs_uv"],
                                                                  n, d["plane"][:, 3], d["intr"])
    assert rc == 0 and term.startswith("CONVERGENCE") and np.array_equal(valid, d["valid"])
    assert trace[-1]["cost"] < 0.02 * trace[0]["cost"]
    assert np.abs(t - d["t_gt"]).max() < np.abs(d["t"] - d["t_gt"]).max()
    L = pkg._lib
    with pytest.raises(L.LvbaError):      # camera index out of range
        pkg.VisualProblem(4, d["obs_off"], d["obs_cam"], d["obs_uv"], d["plane"], d["valid"], d["intr"])
    # max_iter = 0: only the initial evaluation row, state unchanged (quaternions re-normalised on write-back)
    prob = pkg.VisualProblem(8, d["obs_off"], d["obs_cam"], d["obs_uv"], d["plane"], d["valid"], d["intr"])
    (q0, t0, X0), tr0, term0, _ = prob.refine(d["q"], d["t"], d["X"], max_iter=0)
    assert len(tr0) == 1 and term0 == "NO_CONVERGENCE" and np.allclose(q0, d["q"], atol=1e-15) and np.array_equal(t0, d["t"])
    # a point behind its camera contributes a zero residual with zero Jacobian (include/utils.hpp:78)
    Xb = d["X"].copy()
    first = int(np.nonzero(d["valid"])[0][0])
    Xb[first] = d["X"][first] - 1e3 * (d["X"][first] - 0)      # far away on the other side
    from oracle import visual_oracle as vo
    orc = vo.VisualOracle(vo.VisualProblem(d["q"], d["t"], Xb, d["obs_off"], d["obs_cam"], d["obs_uv"], d["plane"], d["valid"], d["intr"]))
    assert abs(prob.cost(d["q"], d["t"], Xb) - orc.cost(*orc.state())) <= 1e-10 * orc.cost(*orc.state())


@pytest.mark.parametrize("seed,track_len", [(3, 5), (4, 4), (7, 8)])
def test_triangulate_tracks_matches_oracle(pkg, synth, seed, track_len):
    """lvba_triangulate_tracks (one lane per track: DLT + 4x4 Jacobi eigen-solver + mean reprojection error) against the
    restated TriangulateTrackDLT / ComputeMeanReproj; includes tracks that must fail (too short, behind a camera)."""
    from importlib import import_module
    from oracle import track_oracle as to
    vis = import_module("global-lvba_amd.visual")
    d = synth.make_visual_problem(10, 200, seed=seed, track_len=track_len)
    q = d["q_gt"]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    Rcw = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                    2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                    2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    off, cam, uv = d["obs_off"].copy(), d["obs_cam"].copy(), d["obs_uv"].copy()
    # cut two tracks below four observations by re-slicing the CSR, and corrupt one observation of another
    keep = np.ones(len(cam), bool)
    keep[off[1]:off[2] - 3] = False if off[2] - off[1] > 3 else True
    cnts = np.diff(off)
    cnts[1] = keep[off[1]:off[2]].sum()
    cam, uv = cam[keep], uv[keep]
    off = np.concatenate([[0], np.cumsum(cnts)])
    uv[off[5]] = [np.nan, 10.0]
    ok_r, X_r, err_r, cnt_r = to.triangulate_tracks(d["intr"], Rcw, d["t_gt"], off, cam, uv)
    ok, X, err, cnt = vis.triangulate_tracks(Rcw, d["t_gt"], off, cam, uv, d["intr"])
    np.testing.assert_array_equal(ok, ok_r)
    np.testing.assert_array_equal(cnt, cnt_r)
    good = ok_r.astype(bool)
    assert good.sum() > 150 and (~good).sum() >= 1
    # the smallest eigenvector of A^T A is conditioned by the gap to the next eigenvalue: compare through the pixel fit
    # and relative to the point's distance
    rel = np.linalg.norm(X[good] - X_r[good], axis=1) / np.linalg.norm(X_r[good], axis=1)
    assert rel.max() < 1e-7
    assert np.abs(err[good] - err_r[good]).max() < 1e-6
    assert np.isinf(err[~good]).all() or (cnt[~good] >= 0).all()


def test_triangulation_golden_fixture(pkg):
    import os
    from importlib import import_module
    vis = import_module("global-lvba_amd.visual")
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tracks_small.npz"))
    ok, X, err, cnt = vis.triangulate_tracks(z["Rcw"], z["tcw"], z["obs_off"], z["obs_cam"], z["obs_uv"], z["intr"])
    np.testing.assert_array_equal(ok, z["ok"])
    np.testing.assert_array_equal(cnt, z["count"])
    assert (np.linalg.norm(X - z["X"], axis=1) / np.linalg.norm(z["X"], axis=1)).max() < 1e-7
    assert np.abs(err - z["mean_reproj"]).max() < 1e-6


def test_matches_to_landmarks_pipeline(pkg, synth):
    """keypoints + pairwise matches -> build_tracks (host BFS) -> seed DLT, view-angle filter, final DLT (GPU): the tracks
    of the synthetic problem come back, and the accepted landmarks sit on the planted ones."""
    from importlib import import_module
    vis = import_module("global-lvba_amd.visual")
    d = synth.make_visual_problem(12, 150, seed=9, track_len=6)
    M = 12
    off, cam, uv = d["obs_off"], d["obs_cam"], d["obs_uv"]
    # keypoint lists per image (shuffled so that keypoint ids are not track ids) and the matches between consecutive
    # observations of every track
    rng = np.random.default_rng(1)
    per_img = [[] for _ in range(M)]
    for o, c in enumerate(cam):
        per_img[c].append(o)
    kp_of_obs = np.zeros(len(cam), np.int64)
    keypoints = []
    for c in range(M):
        order = rng.permutation(len(per_img[c]))
        ids = np.asarray(per_img[c], np.int64)[order]
        kp_of_obs[ids] = np.arange(len(ids))
        keypoints.append(uv[ids])
    pair_m = {}
    for t in range(len(off) - 1):
        obs = list(range(off[t], off[t + 1]))
        for a, b in zip(obs[:-1], obs[1:]):
            i, j = int(cam[a]), int(cam[b])
            ka, kb = int(kp_of_obs[a]), int(kp_of_obs[b])
            if i == j:
                continue
            if i > j:
                i, j, ka, kb = j, i, kb, ka
            pair_m.setdefault((i, j), []).append((ka, kb))
    pairs = sorted(pair_m)
    matches = [np.asarray(pair_m[p]) for p in pairs]
    toff, timg, tkp = vis.build_tracks([len(k) for k in keypoints], pairs, matches, obser_thr=3)
    assert len(toff) - 1 == len(off) - 1                       # every planted track is one component
    q = d["q_gt"]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    Rcw = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                    2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                    2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    ok, X, err, koff, kimg, kkp = vis.triangulate_and_filter(Rcw, d["t_gt"], keypoints, toff, timg, tkp, d["intr"],
                                                             min_view_angle_deg=0.5)
    assert ok.mean() > 0.5 and (err[ok] <= 3.0).all()
    # match recovered tracks to planted landmarks through their first observation
    first_obs = {(int(cam[off[t]]), int(kp_of_obs[off[t]])): t for t in range(len(off) - 1)}
    dists = []
    for t in np.nonzero(ok)[0]:
        key = (int(timg[toff[t]]), int(tkp[toff[t]]))
        if key in first_obs:
            dists.append(np.linalg.norm(X[t] - d["X_gt"][first_obs[key]]))
    assert len(dists) > 50 and np.median(dists) < 0.5


@pytest.mark.parametrize("n_cams,n_tracks,track_len", [(300, 6000, 4), (160, 4000, 9), (1000, 30000, 4)])
def test_block_cyclic_reduction_equals_band_ldlt(pkg, synth, monkeypatch, n_cams, n_tracks, track_len):
    """Narrow reduced camera systems are solved by block cyclic reduction (csrc/bcr.hip: camera half-bandwidth <= 10; block rows
    of 32 scalars for track_len 4, of 64 for track_len 9) instead of the blocked band LDL^T.  Both are direct solvers of the
    same positive definite system: the whole LM trace and the refined cameras must agree to rounding."""
    d = synth.make_visual_problem(n_cams, n_tracks, track_len=track_len, seed=21)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("LVBA_BCR", mode)
        vp = pkg.VisualProblem(n_cams, d["obs_off"], d["obs_cam"], d["obs_uv"], d["plane"], d["valid"], d["intr"])
        info = vp.info()
        assert info["use_band"] == 1 and info["band_blocks"] == track_len - 1
        out[mode] = vp.refine(d["q"], d["t"], d["X"])
        vp.close()
    (q1, t1, X1), tr1, term1, rc1 = out["1"]
    (q0, t0, X0), tr0, term0, rc0 = out["0"]
    assert rc1 == rc0 == 0 and term1 == term0 and len(tr1) == len(tr0) and len(tr1) >= 4
    # the system is ill-conditioned (the constant camera's block carries only the 1e-10 LM diagonal: cond ~ 3e10), so two
    # direct solvers' steps differ by ~cond * eps ~ 3e-6 in the worst direction: the costs along the trace agree to a few 1e-8
    # (3.1e-8 measured at one iteration of the 1 000-camera case), the variables correspondingly less
    for a, b in zip(tr1, tr0):
        assert a["accepted"] == b["accepted"] and abs(a["cost"] - b["cost"]) <= 1e-7 * b["cost"]
    assert np.abs(q1 - q0).max() <= 1e-7 and np.abs(t1 - t0).max() <= 1e-6 and np.abs(X1 - X0).max() <= 1e-5
    assert tr1[-1]["cost"] < 0.1 * tr1[0]["cost"]


def test_reduced_camera_system_at_c3_scale_matches_reference_functors(pkg, synth, monkeypatch):
    """The visual stage AT THE BASELINE.json SCALE (2 000 cameras x 125 000 landmarks x 500 k observations) against the
    reference's own cost functors differentiated with Jets (oracle/_ref, ref_visual_reduced_system: residuals and ambient
    Jacobians are the reference's arithmetic; manifold, Jacobi scaling, LM diagonal and Schur complement restated in C++):
    cost, every block of the reduced camera system S inside its band, nothing outside it, the reduced right-hand side -- and the
    cameras after the FIRST LM iteration, with the reduced system solved by block cyclic reduction (default) and by the band
    LDL^T (LVBA_BCR=0), against a LAPACK banded solve of the reference-functor system."""
    import oracle
    from scipy.linalg import solveh_banded
    from oracle import visual_oracle as vo
    if not oracle.Reference.available():
        pytest.skip("oracle/_ref/libbalm_ref.so absent")
    ref = oracle.Reference()
    M, T = 2000, 125_000
    d = synth.make_visual_problem(M, T, device="cuda")
    args = (d["q"], d["t"], d["X"])
    radius, kb = 1e4, 3
    Sb, rhs_r, c_r, far, sc = ref.visual_reduced_system(*args, d["obs_off"], d["obs_cam"], d["obs_uv"], d["plane"], d["valid"], d["intr"],
                                                       radius=radius, kb=kb)
    assert far <= kb
    # the first step of the reference-functor system: S x = rhs (cameras 1 .. M-1), step = -scale . x, manifold Plus
    n = 6 * (M - 1)
    bwid = 6 * kb + 5
    ab = np.zeros((bwid + 1, n))                             # LAPACK lower band storage: ab[i - j, j] = S[i, j]
    for a in range(1, M):
        for dd in range(min(kb, a - 1) + 1):
            blk = Sb[a, dd]
            for r in range(6):
                for c in range(6):
                    i, j = 6 * (a - 1) + r, 6 * (a - 1 - dd) + c
                    if i >= j:
                        ab[i - j, j] = blk[r, c]
    x = solveh_banded(ab, rhs_r[6:], lower=True)
    step = -(sc[6:] * x).reshape(M - 1, 6)
    q_ref, t_ref = d["q"].copy(), d["t"].copy()
    for cidx in range(1, M):
        q_ref[cidx] = vo.eigen_quat_plus(d["q"][cidx], step[cidx - 1, :3])
        t_ref[cidx] = d["t"][cidx] + step[cidx - 1, 3:]
    q_ref /= np.linalg.norm(q_ref, axis=1, keepdims=True)   # the write-back normalises (src/lvba_system.cpp:1653)
    for bcr in ("1", "0"):
        monkeypatch.setenv("LVBA_BCR", bcr)
        prob = pkg.VisualProblem(M, d["obs_off"], d["obs_cam"], d["obs_uv"], d["plane"], d["valid"], d["intr"])
        if bcr == "1":
            S, rhs, c = prob.linearize(*args, radius)
            assert abs(c - c_r) <= 1e-11 * c_r
            assert np.array_equal(S, S.T)
            worst = 0.0
            Sv = S.reshape(M, 6, M, 6)
            scale_S = np.abs(Sb).max()
            for dd in range(kb + 1):
                a = np.arange(max(1, dd + 1), M)
                got = Sv[a, :, a - dd, :]                    # [len, 6, 6]
                worst = max(worst, float(np.abs(got - Sb[a, dd]).max()) / scale_S)
                Sv[a, :, a - dd, :] = 0.0
                if dd:
                    Sv[a - dd, :, a, :] = 0.0
            assert worst <= 1e-9, worst
            assert np.abs(S[6:, 6:]).max() == 0.0           # nothing outside the band
            assert np.abs(rhs[6:] - rhs_r[6:]).max() <= 1e-9 * np.abs(rhs_r).max()
            del S, Sv
        (q1, t1, X1), trace, term, rc = prob.refine(*args, max_iter=1)
        assert rc == 0 and len(trace) == 2 and trace[1]["accepted"] == 1
        assert np.abs(t1 - t_ref).max() <= 1e-8 and np.abs(q1 - q_ref).max() <= 1e-9, (bcr, np.abs(t1 - t_ref).max())
        prob.close()
