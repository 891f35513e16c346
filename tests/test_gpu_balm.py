"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on identical inputs.

Tolerances: the path is fp64; lambda_min (~1e-4) is a cancellation of second moments (~1e4), so two
correct fp64 implementations agree to ~1e-9 relative per voxel.  north_star's bar is 1e-5 relative on
per-iteration cost and final poses; these tests hold 1e-7 or tighter.
"""
import os

import numpy as np
import pytest

from conftest import make_problem, rel

pytestmark = pytest.mark.gpu

CASES = [
    dict(n_poses=12, n_voxels=60, band=4, seed=1),
    dict(n_poses=20, n_voxels=500, band=20, seed=5),            # window-BA shape: every pose sees every voxel region
    dict(n_poses=40, n_voxels=3000, band=10, seed=2),
    dict(n_poses=100, n_voxels=20000, seed=3),
    dict(n_poses=150, n_voxels=8000, band=12, seed=4),          # banded -> band LDL^T path
]


def _mk(pkg, oracle_mod, case, **kw):
    d = make_problem(**case)
    prob = pkg.BalmProblem(d["n_poses"], d["voxel_off"], d["pose_idx"], d["clusters"], **kw)
    co = oracle_mod.COracle(d["n_poses"], d["voxel_off"], d["pose_idx"], d["clusters"])
    return d, prob, co


@pytest.mark.parametrize("case", CASES)
def test_cost_matches_oracle(pkg, oracle_mod, case):
    d, prob, co = _mk(pkg, oracle_mod, case)
    for x in (d["poses_init"], d["poses_gt"]):
        c_gpu = prob.cost(x)
        c_ref = co.cost(x)
        assert abs(c_gpu - c_ref) <= 1e-8 * abs(c_ref)
        assert abs(prob.cost(x, is_avg=True) - c_ref / prob.n_voxels) <= 1e-8 * abs(c_ref / prob.n_voxels)


@pytest.mark.parametrize("case", CASES)
def test_eval_matches_oracle(pkg, oracle_mod, case):
    d, prob, co = _mk(pkg, oracle_mod, case)
    x = d["poses_init"]
    H, g, c = prob.eval(x)
    Hc, gc, cc = co.eval_dense(x)
    assert abs(c - cc) <= 1e-8 * abs(cc)
    assert rel(g, gc) <= 1e-8
    assert rel(H, Hc) <= 1e-8
    assert np.array_equal(H, H.T)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("u", [0.01, 10.0])
def test_solve_matches_oracle(pkg, oracle_mod, case, u):
    d, prob, co = _mk(pkg, oracle_mod, case)
    x = d["poses_init"]
    prob.eval(x, want_H=False, want_g=False)
    dx = prob.solve(u)
    Hc, gc, _ = co.eval_dense(x)
    A = Hc + u * np.diag(np.diag(Hc))
    dx_ref, rc = oracle_mod.ldlt_solve_dense(A, -gc)
    assert rc == 0
    # residual of the damped system and agreement with the oracle's unpivoted LDL^T
    assert np.abs(A @ dx + gc).max() <= 1e-8 * np.abs(gc).max()
    assert rel(dx, dx_ref) <= 1e-6


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("solver", ["auto", "dense", "natural"])
def test_refine_trace_matches_oracle(pkg, oracle_mod, case, solver):
    kw = {"auto": {}, "dense": dict(band_frac=0.0), "natural": dict(ordering=0)}[solver]
    d, prob, co = _mk(pkg, oracle_mod, case, **kw)
    x_gpu, trace, rc = prob.refine(d["poses_init"])
    x_ref, tr_ref, rc_ref = co.damping_iter(d["poses_init"])
    assert rc == 0 and rc_ref == 0
    _compare_traces(trace, tr_ref, x_gpu, x_ref)


def _compare_traces(trace, tr_ref, x_gpu, x_ref, tol=1e-7):
    """Row-by-row LM trace equality.  Once |q| = |residual1 - residual2| falls to the fp64 noise floor of
    the cost (~1e-8 relative: lambda_min is a 1e8:1 cancellation) the accept/reject branch is a coin flip
    between two correct implementations ('Determinism'), so from the first such row on
    only the converged cost and poses are compared."""
    tie = None
    for i, (row, ref) in enumerate(zip(trace, tr_ref)):
        if abs(ref[5]) <= 3e-8 * abs(ref[1]) or abs(row["q"]) <= 3e-8 * abs(row["residual1"]):
            tie = i
            break
        assert row["accepted"] == int(ref[7]) and row["evaluated"] == int(ref[8])
        assert abs(row["residual1"] - ref[1]) <= tol * abs(ref[1])
        assert abs(row["residual2"] - ref[2]) <= tol * abs(ref[2])
        assert abs(row["u"] - ref[3]) <= 1e-4 * abs(ref[3])
    if tie is None:
        assert len(trace) == len(tr_ref)
        assert np.abs(x_gpu - x_ref).max() <= 1e-7
    else:
        best = min(r["residual2"] if r["accepted"] else r["residual1"] for r in trace)
        best_ref = min(r[2] if r[7] else r[1] for r in tr_ref)
        assert abs(best - best_ref) <= 1e-7 * best_ref
        assert np.abs(x_gpu - x_ref).max() <= 1e-5   # north_star's bar; the coin-flip step itself is ~1e-6


@pytest.mark.parametrize("name", ["balm_small", "balm_window", "balm_reject"])
def test_hip_matches_reference_golden(pkg, name):
    """The HIP path against what the REFERENCE'S OWN CODE answers on the committed fixtures (tests/golden/ref_balm.npz:
    BALM2::divide_thread and BALM2::damping_iter of include/BALM/bavoxel.hpp, compiled from the reference's sources against
    the stand-ins of oracle/shim by tests/golden/make_golden.py:main_ref).  No oracle in between."""
    import os
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z, r = np.load(os.path.join(gd, name + ".npz")), np.load(os.path.join(gd, "ref_balm.npz"))
    prob = pkg.BalmProblem(int(z["n_poses"]), z["voxel_off"], z["pose_idx"], z["clusters"])
    x0 = z["poses_init"]
    H, g, c = prob.eval(x0)
    assert abs(c - r[name + "__cost_avg"]) <= 1e-8 * r[name + "__cost_avg"]
    assert rel(g, r[name + "__g"]) <= 1e-8 and rel(H, r[name + "__H"]) <= 1e-8
    assert abs(prob.cost(x0) - r[name + "__cost_sum"]) <= 1e-8 * r[name + "__cost_sum"]
    xf, trace, rc = prob.refine(x0)
    assert rc == 0
    assert np.abs(xf - r[name + "__poses_final"]).max() <= 1e-5              # BASELINE.json: refined poses within 1e-5
    cf = prob.cost(xf, is_avg=True)
    assert abs(cf - r[name + "__cost_final_avg"]) <= 1e-5 * r[name + "__cost_final_avg"]   # and the converged cost


def test_refine_reject_branch(pkg, oracle_mod):
    """A start far enough from the optimum that the first steps are rejected (u *= v; v *= 2;
    Hessian reused, bavoxel.hpp:753-758)."""
    case = dict(n_poses=12, n_voxels=60, band=4, seed=1, rot_sigma_deg=0.03, trans_sigma=0.02)
    d, prob, co = _mk(pkg, oracle_mod, case)
    x_gpu, trace, rc = prob.refine(d["poses_init"])
    x_ref, tr_ref, _ = co.damping_iter(d["poses_init"])
    assert any(not r["accepted"] for r in trace)
    assert [r["accepted"] for r in trace] == [int(r[7]) for r in tr_ref]
    assert [r["evaluated"] for r in trace] == [int(r[8]) for r in tr_ref]
    for row, ref in zip(trace, tr_ref):
        assert abs(row["residual2"] - ref[2]) <= 1e-6 * abs(ref[2])
    assert np.abs(x_gpu - x_ref).max() <= 1e-6


def test_step_api_equals_refine(pkg, oracle_mod):
    d, prob, _ = _mk(pkg, oracle_mod, CASES[2])
    x1, trace, _ = prob.refine(d["poses_init"])
    prob.lm_begin(d["poses_init"])
    rows, done = [], False
    while not done:
        row, done, rc = prob.lm_step()
        rows.append(row)
    x2 = prob.lm_end()
    assert len(rows) == len(trace)
    assert np.array_equal(x1, x2)          # no atomics anywhere on the path: bitwise reproducible
    with pytest.raises(pkg._lib.LvbaError):
        prob.lm_step()


def test_reference_interface_mirror(pkg, oracle_mod):
    """VOX_HESS.push_voxel / BALM2.divide_thread / only_residual / damping_iter, called the way
    src/lvba_system.cpp:253-264 calls the reference."""
    d = make_problem(12, 60, band=4, seed=1)
    N = d["n_poses"]
    voxhess = pkg.VOX_HESS(N)
    off, idx, clu = d["voxel_off"], d["pose_idx"], d["clusters"]
    for a in range(len(off) - 1):
        sig = np.zeros((N, 10))
        sig[idx[off[a]:off[a + 1]]] = clu[off[a]:off[a + 1]]
        voxhess.push_voxel(sig)
    lone = np.zeros((N, 10)); lone[3] = clu[0]
    voxhess.push_voxel(lone)                                   # < 2 observers: not admitted (bavoxel.hpp:52)
    assert len(voxhess.plvec_voxels) == len(off) - 1
    x_stats = [pkg.IMUST(row[:9].reshape(3, 3), row[9:]) for row in d["poses_init"]]
    opt = pkg.BALM2(N)
    co = oracle_mod.COracle(N, off, idx, clu)
    r, H, g = opt.divide_thread(x_stats, voxhess)
    Hc, gc, cc = co.eval_dense(d["poses_init"])
    assert abs(r - cc) <= 1e-8 * cc and rel(H, Hc) <= 1e-8 and rel(g, gc) <= 1e-8
    assert abs(opt.only_residual(x_stats, voxhess, is_avg=True) - cc) <= 1e-8 * cc
    opt.damping_iter(x_stats, voxhess)
    x_ref, tr_ref, _ = co.damping_iter(d["poses_init"])
    got = np.stack([np.concatenate([x.R.reshape(9), x.p]) for x in x_stats])
    assert np.abs(got - x_ref).max() <= 1e-7
    assert len(opt.last_trace) == len(tr_ref)


def test_edge_cases(pkg, oracle_mod):
    L = pkg._lib
    d = make_problem(12, 60, band=4, seed=1)
    # a voxel with a single factor is refused (push_voxel admission rule)
    with pytest.raises(L.LvbaError) as e:
        pkg.BalmProblem(12, np.array([0, 1, 3]), d["pose_idx"][:3], d["clusters"][:3])
    assert e.value.code == L.ERR_ARG
    # pose index out of range
    bad = d["pose_idx"].copy(); bad[0] = 99
    with pytest.raises(L.LvbaError):
        pkg.BalmProblem(12, d["voxel_off"], bad, d["clusters"])
    # non-zero CSR base (a shard handed over with global offsets)
    off = d["voxel_off"]
    lo = 20
    sub = pkg.BalmProblem(12, off[lo:], d["pose_idx"][off[lo]:], d["clusters"][off[lo]:])
    co = oracle_mod.COracle(12, off[lo:] - off[lo], d["pose_idx"][off[lo]:], d["clusters"][off[lo]:])
    assert abs(sub.cost(d["poses_init"]) - co.cost(d["poses_init"])) <= 1e-8 * co.cost(d["poses_init"])
    # max_iter = 0 leaves the poses untouched; solve before eval is a state error
    p = pkg.BalmProblem(12, off, d["pose_idx"], d["clusters"])
    with pytest.raises(L.LvbaError) as e:
        p.solve(0.01)
    assert e.value.code == L.ERR_STATE
    x, trace, rc = p.refine(d["poses_init"], max_iter=0)
    assert trace == [] and np.array_equal(x, d["poses_init"])
    # a pose observed by no voxel: zero pivot -> numerical status, like an unchecked LDLT failure upstream
    idx2 = d["pose_idx"].copy()
    d13 = pkg.BalmProblem(13, off, idx2, d["clusters"])
    x13 = np.concatenate([d["poses_init"], d["poses_init"][-1:]])
    _, _, rc = d13.refine(x13)
    assert rc == L.NUM_FACTORIZATION


def test_max_observers_per_voxel(pkg, oracle_mod):
    """One voxel seen by many poses (k = 40) next to ordinary ones: exercises the pair enumeration."""
    d = make_problem(60, 400, band=29, k_max=40, k_extra_mean=25.0, seed=7)
    assert np.diff(d["voxel_off"]).max() >= 30
    prob = pkg.BalmProblem(d["n_poses"], d["voxel_off"], d["pose_idx"], d["clusters"])
    co = oracle_mod.COracle(d["n_poses"], d["voxel_off"], d["pose_idx"], d["clusters"])
    H, g, c = prob.eval(d["poses_init"])
    Hc, gc, cc = co.eval_dense(d["poses_init"])
    assert rel(H, Hc) <= 1e-8 and rel(g, gc) <= 1e-8 and abs(c - cc) <= 1e-8 * cc


def test_properties_at_scale(pkg, synth):
    """Size-independent properties at a size the dense oracle cannot reach (N=500, ~250k factors):
    rigid-motion invariance of the cost, gradient = directional derivative of the cost-only kernel,
    Hessian-vector product = directional derivative of the gradient, linearity over voxel shards."""
    d = make_problem(500, 50000, seed=11)
    N = d["n_poses"]
    prob = pkg.BalmProblem(N, d["voxel_off"], d["pose_idx"], d["clusters"])
    x = d["poses_init"]
    c0 = prob.cost(x)
    # (1) left-multiplying every pose by one rigid motion leaves every lambda_min unchanged
    th = 0.7
    Rg = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    tg = np.array([3.0, -2.0, 0.5])
    R = x[:, :9].reshape(-1, 3, 3); p = x[:, 9:]
    xg = np.concatenate([(Rg @ R).reshape(-1, 9), p @ Rg.T + tg], axis=1)
    assert abs(prob.cost(xg) - c0) <= 1e-7 * c0
    # (2)/(3) directional derivatives through the retraction, using the oracle-free GPU kernels only
    import oracle.balm_oracle as bo
    H, g, _ = prob.eval(x)
    rng = np.random.default_rng(0)
    dvec = rng.standard_normal(6 * N)
    h = 1e-6
    cp, cm = prob.cost(bo.retract(x, h * dvec)), prob.cost(bo.retract(x, -h * dvec))
    assert abs((cp - cm) / (2 * h) - g @ dvec) <= 1e-5 * abs(g @ dvec)
    # (4) shards add up: cost, g and H are sums over voxel ranges (bavoxel.hpp:626-633)
    off = d["voxel_off"]
    Hs, gs, cs = 0, 0, 0
    for r in range(3):
        a, b = pkg.shard_range(len(off) - 1, r, 3)
        sh = pkg.BalmProblem(N, off[a:b + 1], d["pose_idx"][off[a]:off[b]], d["clusters"][off[a]:off[b]])
        Hr, gr, cr = sh.eval(x)
        Hs, gs, cs = Hs + Hr, gs + gr, cs + cr * (b - a)
        sh.close()
    assert rel(Hs, H) <= 1e-10 and rel(gs, g) <= 1e-10
    assert abs(cs - c0) <= 1e-10 * c0
    # LM at this size: monotone accepted costs, ends below the ground-truth cost
    xf, trace, rc = prob.refine(x)
    assert rc == 0
    acc = [r for r in trace if r["accepted"]]
    assert len(acc) >= 2 and all(r["residual2"] < r["residual1"] for r in acc)
    assert acc[-1]["residual2"] <= prob.cost(d["poses_gt"], is_avg=True) * 1.001


def test_rccl_path_single_rank(pkg, oracle_mod, monkeypatch):
    """The RCCL plumbing (dlopen, unique id, communicator, all-reduce of {H, g, cost} and of the cost
    scalar) exercised with a 1-rank communicator -- all a 1-GPU box can run; the multi-rank arithmetic is
    covered by tests/test_dist_cpu.py."""
    monkeypatch.setenv("LVBA_SINGLE_RANK_COMM", "1")
    d = make_problem(40, 3000, band=10, seed=2)
    prob = pkg.BalmProblem(d["n_poses"], d["voxel_off"], d["pose_idx"], d["clusters"])
    uid = pkg.BalmProblem.unique_id()
    assert len(uid) == 128
    prob.dist_init(1, 0, uid)
    co = oracle_mod.COracle(d["n_poses"], d["voxel_off"], d["pose_idx"], d["clusters"])
    H, g, c = prob.eval(d["poses_init"])
    Hc, gc, cc = co.eval_dense(d["poses_init"])
    assert rel(H, Hc) <= 1e-8 and rel(g, gc) <= 1e-8 and abs(c - cc) <= 1e-8 * cc
    H2, g2, c2 = prob.eval(d["poses_init"])                 # the store is re-zeroed before every reduce
    assert np.array_equal(H, H2) and np.array_equal(g, g2)
    x, trace, rc = prob.refine(d["poses_init"])
    xr, tr, _ = co.damping_iter(d["poses_init"])
    assert rc == 0 and np.abs(x - xr).max() <= 1e-7
    assert prob.info()["n_voxels_global"] == 3000


def test_packed_allreduce_single_rank(pkg, oracle_mod, monkeypatch):
    """With a communicator and a system large enough for the ordering to be computed (n > 1024), only the blocks of the union
    sparsity pattern are all-reduced (bs_allreduce_hg: pack -> RCCL -> unpack; a pattern that fills more than 3/4 of the
    block-band store travels whole).  1-rank communicator: the result must be bitwise what the handle without a communicator
    gives, and match the oracle."""
    monkeypatch.setenv("LVBA_SINGLE_RANK_COMM", "1")
    d = make_problem(200, 6000, band=10, seed=7)
    out = {}
    for mode in ("comm", "plain"):
        prob = pkg.BalmProblem(d["n_poses"], d["voxel_off"], d["pose_idx"], d["clusters"])
        if mode == "comm":
            prob.dist_init(1, 0, pkg.BalmProblem.unique_id())
        H, g, c = prob.eval(d["poses_init"])
        H2, g2, c2 = prob.eval(d["poses_gt"])
        x, trace, rc = prob.refine(d["poses_init"])
        assert rc == 0
        info = prob.info()
        if mode == "comm":   # the packed buffer really is smaller than the block-band store
            assert 0 < info["allreduce_bytes"] < 0.75 * info["hess_bytes"]
        else:
            assert info["allreduce_bytes"] == 0
        out[mode] = (H, g, c, H2, g2, c2, x)
        prob.close()
    for a, b in zip(out["comm"], out["plain"]):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    co = oracle_mod.COracle(d["n_poses"], d["voxel_off"], d["pose_idx"], d["clusters"])
    Hc, gc, cc = co.eval_dense(d["poses_init"])
    assert rel(out["comm"][0], Hc) <= 1e-8 and rel(out["comm"][1], gc) <= 1e-8 and abs(out["comm"][2] - cc) <= 1e-8 * cc

def test_properties_at_baseline_size_c2(pkg, synth):
    """BASELINE.json config C2 (500 poses x 400k voxels x 2M factors), generated on the GPU: size-independent
    properties -- rigid-motion invariance, shard additivity of the cost, bitwise run-to-run reproducibility of the
    atomic-free assembly, gradient = directional derivative of the cost kernel, and a full refine that ends at the
    ground-truth cost level."""
    import torch
    N, V = synth.CONFIGS["C2"]
    d = synth.make_balm_problem(N, V, device="cuda")
    torch.cuda.empty_cache()
    prob = pkg.BalmProblem(N, d["voxel_off"], d["pose_idx"], d["clusters"])
    x = d["poses_init"]
    c0 = prob.cost(x)
    th = -0.4
    Rg = np.array([[1, 0, 0], [0, np.cos(th), -np.sin(th)], [0, np.sin(th), np.cos(th)]])
    tg = np.array([-7.0, 2.0, 1.5])
    xg = np.concatenate([(Rg @ x[:, :9].reshape(-1, 3, 3)).reshape(-1, 9), x[:, 9:] @ Rg.T + tg], axis=1)
    assert abs(prob.cost(xg) - c0) <= 1e-7 * c0
    _, g, _ = prob.eval(x, want_H=False)
    _, g2, _ = prob.eval(x, want_H=False)
    assert np.array_equal(g, g2)                                        # deterministic assembly
    import oracle.balm_oracle as bo
    dv = np.random.default_rng(1).standard_normal(6 * N)
    h = 2e-6
    fd = (prob.cost(bo.retract(x, h * dv)) - prob.cost(bo.retract(x, -h * dv))) / (2 * h)
    assert abs(fd - g @ dv) <= 1e-5 * np.linalg.norm(g) * np.linalg.norm(dv)
    off = d["voxel_off"]
    cs = 0.0
    for r in range(2):
        a, b = pkg.shard_range(V, r, 2)
        sh = pkg.BalmProblem(N, off[a:b + 1], d["pose_idx"][off[a]:off[b]], d["clusters"][off[a]:off[b]])
        cs += sh.cost(x)
        sh.close()
    assert abs(cs - c0) <= 1e-12 * c0
    xf, trace, rc = prob.refine(x)
    assert rc == 0 and trace[-1]["residual2"] <= prob.cost(d["poses_gt"], is_avg=True) * 1.001


def test_voxels_with_more_observers_than_lanes(pkg, oracle_mod):
    """Voxels seen by more poses than a workgroup has lanes (> 256): merged in tiles by a workgroup of their own.  Cost, H
    and g against the C oracle on a problem that mixes them with ordinary voxels (first, middle and last position)."""
    rng = np.random.default_rng(7)
    N = 320
    d = make_problem(N, 500, band=12, seed=5)
    off, idx, clu = d["voxel_off"], d["pose_idx"], d["clusters"].reshape(-1, 10)
    t = np.arange(N) * (2 * np.pi / N)

    def big_voxel(k):
        # a horizontal plane patch far below the trajectory, seen by k poses: points in the world frame, moved to each body
        poses = rng.choice(N, k, replace=False)
        poses.sort()
        out = []
        for p in poses:
            T = d["poses_gt"][p]
            R, tr = T[:9].reshape(3, 3), T[9:]
            pw = np.column_stack([rng.uniform(-0.4, 0.4, (20, 2)) + [3.0, -2.0], np.full(20, -30.0) + rng.normal(0, 0.01, 20)])
            pb = ((pw - tr) @ R).astype(np.float32).astype(np.float64)
            out.append(np.concatenate([[np.sum(pb[:, 0] * pb[:, 0]), np.sum(pb[:, 0] * pb[:, 1]), np.sum(pb[:, 0] * pb[:, 2]),
                                        np.sum(pb[:, 1] * pb[:, 1]), np.sum(pb[:, 1] * pb[:, 2]), np.sum(pb[:, 2] * pb[:, 2])],
                                       pb.sum(0), [20.0]]))
        return poses.astype(np.int32), np.asarray(out)
    bigs = [big_voxel(k) for k in (300, 257, 320)]
    V = len(off) - 1
    pieces = [(bigs[0][0], bigs[0][1])]
    for v in range(V):
        pieces.append((idx[off[v]:off[v + 1]], clu[off[v]:off[v + 1]]))
        if v == V // 2:
            pieces.append(bigs[1])
    pieces.append(bigs[2])
    new_off = np.concatenate([[0], np.cumsum([len(p[0]) for p in pieces])]).astype(np.int64)
    new_idx = np.concatenate([p[0] for p in pieces]).astype(np.int32)
    new_clu = np.concatenate([p[1] for p in pieces])
    prob = pkg.BalmProblem(N, new_off, new_idx, new_clu)
    co = oracle_mod.COracle(N, new_off, new_idx, new_clu)
    x0 = d["poses_init"]
    assert abs(prob.cost(x0) - co.cost(x0)) <= 1e-9 * co.cost(x0)
    H, g, c = prob.eval(x0)
    Hc, gc, cc = co.eval_dense(x0)
    assert abs(c - cc) <= 1e-9 * cc
    assert rel(g, gc) <= 1e-8 and rel(H, Hc) <= 1e-8
    x, trace, rc = prob.refine(x0)
    xr, tr, _ = co.damping_iter(x0)
    assert rc == 0 and len(trace) == len(tr) and np.abs(x - xr).max() <= 1e-7


def test_voxel_order_unrelated_to_the_poses(pkg):
    """The passes over the voxels (and above all the voxel windows of the pair lists) are laid out for voxels that come roughly
    in the order of the poses that see them; the synthetic problems and the voxel maps do.  The reference hands its voxels over
    in the iteration order of an unordered_map, i.e. in RANDOM order: lvba_balm_create then re-lays a large problem internally
    (voxels sorted by the first pose that sees them; below the size threshold the pair lists fall back to the plain
    block-major form).  Either way the result is the sum over the same voxels: H, g and the cost must not
    depend on the order beyond rounding, and LM runs agree.  (250 k factors: above the threshold.)"""
    d = make_problem(500, 50000, seed=11)
    off, idx, clu = d["voxel_off"], d["pose_idx"], d["clusters"]
    V = len(off) - 1
    perm = np.random.default_rng(3).permutation(V)
    k = np.diff(off)
    new_off = np.concatenate([[0], np.cumsum(k[perm])]).astype(np.int64)
    gather = np.concatenate([np.arange(off[v], off[v + 1]) for v in perm])
    a = pkg.BalmProblem(d["n_poses"], off, idx, clu)
    b = pkg.BalmProblem(d["n_poses"], new_off, idx[gather], clu[gather])
    Ha, ga, ca = a.eval(d["poses_init"])
    Hb, gb, cb = b.eval(d["poses_init"])
    assert rel(Hb, Ha) <= 1e-12 and rel(gb, ga) <= 1e-12 and abs(cb - ca) <= 1e-12 * ca
    xa, ta, _ = a.refine(d["poses_init"])
    xb, tb, _ = b.refine(d["poses_init"])
    assert len(ta) == len(tb) and np.abs(xa - xb).max() <= 1e-8


_SCHEDULE_SCRIPT = r"""
import importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
pkg = importlib.import_module("global-lvba_amd")
synth = importlib.import_module("global-lvba_amd.synth")
d = synth.make_balm_problem(700, 30000, band=12, loop_frac=0.0, seed=5)
prob = pkg.BalmProblem(700, d["voxel_off"], d["pose_idx"], d["clusters"])
import os
prob.eval(d["poses_init"], want_H=False, want_g=False)
info = prob.info()
dx = prob.solve(0.01)
if os.environ.get("LVBA_CHECK_BAND") == "1":      # the check runs at the START of a solve, on what the solves before it left
    for _ in range(2):
        assert np.array_equal(prob.solve(0.01), dx)
np.save(sys.argv[2], np.concatenate([dx.ravel(), [info["use_band"], info["twist_panels"]]]))
"""


def test_solver_schedules(tmp_path):
    """The band LDL^T (look-ahead schedule: one launch per panel; both ends at once, paired panels, 128 x 64 update tiles) has two
    forms a problem can end up in by its size and shape -- 64 x 64 update tiles with 64-bit addressing (matrices of 4 GB and
    more; LVBA_SOLVER=bulk64 forces it) and the plain top-down factorisation (bands too short for two ends; LVBA_SOLVER=notwist);
    the two ends run the row roles' deferred form, everything else their full form (LVBA_SOLVER=nodefer: the full form everywhere).
    Every form must give the same solution of the same damped system (they differ in summation order only): 4200 unknowns,
    half-bandwidth ~150, enough panels for the two-ended form and the pairing to be active.  (Environment switches are read
    once per process: one subprocess per form.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "run.py"
    script.write_text(_SCHEDULE_SCRIPT)
    variants = [{}, {"LVBA_SOLVER": "bulk64"}, {"LVBA_SOLVER": "notwist"}, {"LVBA_SOLVER": "bulk64,notwist"},
                # the row roles' full form in the two-ended phases too (default there: the deferred form, ldlt_schedule.h)
                {"LVBA_SOLVER": "nodefer"}, {"LVBA_SOLVER": "bulk64,nodefer"},
                # debugging switch: the never-rewritten part of the band store stays zero over repeated solves (no graph)
                {"LVBA_CHECK_BAND": "1", "LVBA_NO_GRAPH": "1"}]
    out = []
    for i, v in enumerate(variants):
        f = tmp_path / f"dx_{i}.npy"
        env = dict(os.environ, **v)
        r = subprocess.run([sys.executable, str(script), root, str(f)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (v, r.stderr[-2000:])
        out.append(np.load(f))
    ref = out[0]
    assert ref[-2] == 1 and ref[-1] >= 4, ref[-2:]          # band storage, factorised from both ends
    assert np.isfinite(ref).all()
    for v, o in zip(variants[1:], out[1:]):
        assert np.abs(o[:-2] - ref[:-2]).max() <= 1e-9 * np.abs(ref[:-2]).max(), (v, np.abs(o[:-2] - ref[:-2]).max() / np.abs(ref[:-2]).max())


def test_grouped_refinement_equals_one_by_one(pkg):
    """Several independent problems in one handle (lvba_balm_set_groups / lvba_balm_refine_groups: the windows of
    LvbaSystem::runWindowBA, src/lvba_system.cpp:232-302) advance through ONE LM loop in lock-step -- one evaluation, one band
    factorisation with a damping value per group, one cost pass per iteration -- while every group keeps its own u, v,
    accept / reject and exit test.  Each group must end where lvba_balm_refine puts it when it is refined alone (the arithmetic
    differs in summation order only), after the same number of iterations; the groups here differ in size, in how far they are
    from the optimum and (max_iter small) in whether they run into the iteration cap."""
    specs = [(20, 3000, 3), (12, 1500, 4), (20, 2500, 5), (8, 900, 6)]
    parts = [make_problem(n, v, seed=s, band=50, loop_frac=0.0) for n, v, s in specs]
    # scale the initial error differently per group so that the LM paths differ
    inits = []
    for i, d in enumerate(parts):
        x0, xg = d["poses_init"].copy(), d["poses_gt"]
        inits.append(xg + (x0 - xg) * (0.3 + 0.9 * i))
    pose_off = np.cumsum([0] + [d["n_poses"] for d in parts]).astype(np.int32)
    vox_off = np.cumsum([0] + [len(d["voxel_off"]) - 1 for d in parts]).astype(np.int64)
    off = [np.zeros(1, np.int64)]
    idx, clu = [], []
    for k, d in enumerate(parts):
        off.append(d["voxel_off"][1:] + off[-1][-1])
        idx.append(d["pose_idx"] + pose_off[k])
        clu.append(d["clusters"])
    off, idx, clu = np.concatenate(off), np.concatenate(idx).astype(np.int32), np.concatenate(clu)
    x0 = np.concatenate(inits)
    for max_iter in (10, 3):
        u = pkg.BalmProblem(int(pose_off[-1]), off, idx, clu)
        u.set_groups(pose_off, vox_off)
        xu, per, rc = u.refine_groups(x0, max_iter=max_iter)
        assert rc == 0
        its = []
        for k, d in enumerate(parts):
            p = pkg.BalmProblem(d["n_poses"], d["voxel_off"], d["pose_idx"], d["clusters"])
            xs, tr, rck = p.refine(inits[k], max_iter=max_iter)
            assert rck == 0
            its.append(len(tr))
            assert per["n_iter"][k] == len(tr), (k, per["n_iter"][k], len(tr))
            assert abs(per["cost_first"][k] - tr[0]["residual1"]) <= 1e-9 * tr[0]["residual1"]
            last = tr[-1]["residual2"] if tr[-1]["accepted"] else tr[-1]["residual1"]
            assert abs(per["cost_last"][k] - last) <= 1e-8 * last
            assert np.abs(xu[pose_off[k]:pose_off[k + 1]] - xs).max() <= 1e-8, k
            p.close()
        if max_iter == 10:
            assert len(set(its)) > 1          # the groups really leave the loop at different iterations
        u.close()
    # a voxel seen from another group's pose is refused
    bad = pkg.BalmProblem(int(pose_off[-1]), off, idx, clu)
    with pytest.raises(Exception):
        bad.set_groups(np.array([0, 10, pose_off[-1]], np.int32), np.array([0, 100, vox_off[-1]], np.int64))
    bad.close()


def test_y32_switch_keeps_cost_gradient_and_lm_trace(pkg, synth, monkeypatch):
    """LVBA_Y32=1 (an experiment of round 4): the per-factor Y records travel as fp32 between the factor and the pair pass.  Only
    the OFF-DIAGONAL pose blocks may move (they are sums of products of the rounded records, ~6e-7 relative); cost, gradient and
    diagonal blocks come from fp64 registers and must not move at all; and what north_star judges -- every LM cost of a
    refinement and the refined poses -- stays far inside its 1e-5 (held here at 1e-8 against the fp64-record run and against
    the oracle's trace).  The switch only applies where the column pair kernel runs (windowed pair lists: LVBA_PAIR_WINDOW forces
    them at this size)."""
    d = synth.make_balm_problem(300, 60000, seed=9)
    x0 = d["poses_init"]
    monkeypatch.setenv("LVBA_PAIR_WINDOW", "4096")
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("LVBA_Y32", mode)
        prob = pkg.BalmProblem(300, d["voxel_off"], d["pose_idx"], d["clusters"])
        assert prob.info()["y_fp32"] == int(mode)
        H, g, c = prob.eval(x0)
        x, trace, rc = prob.refine(x0)
        assert rc == 0
        out[mode] = (H, g, c, x, trace)
        prob.close()
    (H0, g0, c0, xa, ta), (H1, g1, c1, xb, tb) = out["0"], out["1"]
    assert c1 == c0 and np.array_equal(g1, g0)
    n = 6 * 300
    Hb0, Hb1 = H0.reshape(300, 6, 300, 6), H1.reshape(300, 6, 300, 6)
    for i in range(300):
        assert np.array_equal(Hb0[i, :, i, :], Hb1[i, :, i, :])           # diagonal blocks: fp64 registers either way
    assert not np.array_equal(H1, H0) and rel(H1, H0) <= 5e-6               # it did take effect, and stays at fp32 rounding
    assert np.array_equal(H1, H1.T)
    assert len(ta) == len(tb) and [r["accepted"] for r in ta] == [r["accepted"] for r in tb]
    for ra, rb in zip(ta, tb):
        assert abs(ra["residual1"] - rb["residual1"]) <= 1e-8 * abs(ra["residual1"])
        assert abs(ra["residual2"] - rb["residual2"]) <= 1e-8 * abs(ra["residual2"])
    assert np.abs(xa - xb).max() <= 1e-8
    assert n == H0.shape[0]


def test_eval_blocks_is_the_structural_pattern_of_the_dense_hessian(pkg, oracle_mod):
    """lvba_balm_eval_blocks (the sparse export the C4-size parity runs use): the sizing call runs no evaluation and already
    knows the block count (pair-list destinations + the diagonal), the blocks equal those of the dense export entry for entry,
    every non-zero block of the dense matrix is in the list, each unordered pose pair once (bi >= bj)."""
    d, prob, co = _mk(pkg, oracle_mod, CASES[4])
    x = d["poses_init"]
    N = d["n_poses"]
    H, g, c = prob.eval(x)
    bi, bj, blocks, g2, c2 = prob.eval_blocks(x)
    assert c2 == c and np.array_equal(g2, g)
    assert (bi >= bj).all() and len(set(zip(bi.tolist(), bj.tolist()))) == len(bi)
    Hb = H.reshape(N, 6, N, 6)
    seen = np.zeros((N, N), bool)
    for k in range(len(bi)):
        assert np.array_equal(blocks[k], Hb[bi[k], :, bj[k], :]), (bi[k], bj[k])
        seen[bi[k], bj[k]] = True
    nz = np.abs(Hb).max(axis=(1, 3)) > 0
    assert not (np.tril(nz) & ~seen).any()                                  # nothing of H is missing from the list
    assert seen.diagonal().all()
