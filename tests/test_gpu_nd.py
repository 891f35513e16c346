"""The damped solve by one level of nested dissection (csrc/nd_plan.h + csrc/ldlt_nd.h) -- the form the solver takes when the
pose co-visibility graph is not a narrow band: a HUB on the ring (a place crossed many times: synth's revisit="lot"), or a long
band shared by several ranks (chunks of the band ordering as separators).  The reference hands whatever pattern arrives to
Eigen::SimplicialLDLT (include/BALM/bavoxel.hpp:696-710); the checks here are the ones the band solver has: the solution of the
damped system against a dense solve of the SAME system on the host, whole LM runs against the C oracle, and -- multi-rank, as
host threads through tests/host_transport.cpp -- bitwise agreement of the ranks.  LVBA_SOLVER=nd forces the dissection at test
sizes (the cost model would keep two dozen panels in one band); nothing else is switched."""
import numpy as np
import pytest

from conftest import HostTransport, make_problem, rel

pytestmark = pytest.mark.gpu


def dense_solve(H, g, u):
    A = H + u * np.diag(np.diag(H))
    return np.linalg.solve(A, -g)


@pytest.mark.parametrize("case", [dict(n_poses=320, n_voxels=16000, band=12, seed=3, revisit="lot"),
                                  dict(n_poses=420, n_voxels=20000, band=10, seed=5, revisit="lot"),
                                  dict(n_poses=400, n_voxels=20000, band=8, seed=8, revisit="chords")])
def test_hub_graph_dissection_matches_dense_solve_and_oracle(pkg, oracle_mod, case, monkeypatch):
    d = make_problem(**case)
    N = d["n_poses"]
    monkeypatch.setenv("LVBA_SOLVER", "nond")
    band = pkg.BalmProblem(N, d["voxel_off"], d["pose_idx"], d["clusters"])
    Hb, gb, cb = band.eval(d["poses_init"])
    assert band.info()["nd_kind"] == 0
    monkeypatch.setenv("LVBA_SOLVER", "nd")
    prob = pkg.BalmProblem(N, d["voxel_off"], d["pose_idx"], d["clusters"])
    info = prob.info()
    if case["revisit"] == "chords" and info["nd_kind"] == 0:
        pytest.skip("no hub in this graph: the plan kept the band")
    assert info["nd_kind"] == 1 and info["nd_arcs"] >= 1 and 0 < info["nd_sep_poses"] < N // 2, info
    H, g, c = prob.eval(d["poses_init"])
    # the evaluation does not depend on the order the poses are stored in
    assert rel(H, Hb) <= 1e-12 and rel(g, gb) <= 1e-12 and abs(c - cb) <= 1e-12 * cb
    for u in (0.01, 1e-4, 3.0):
        dx = prob.solve(u)
        want = dense_solve(H, g, u)
        assert np.isfinite(dx).all()
        assert np.abs(dx - want).max() <= 1e-8 * np.abs(want).max(), (u, np.abs(dx - want).max() / np.abs(want).max())
        assert np.abs(dx - band.solve(u)).max() <= 1e-8 * np.abs(want).max()
        assert np.array_equal(prob.solve(u), dx)                     # the same launches, the same sums: bitwise
    x, trace, rc = prob.refine(d["poses_init"])
    xb, trb, rcb = band.refine(d["poses_init"])
    co = oracle_mod.COracle(N, d["voxel_off"], d["pose_idx"], d["clusters"])
    xr, tr, _ = co.damping_iter(d["poses_init"])
    assert rc == 0 and rcb == 0 and len(trace) == len(tr) == len(trb)
    assert [r["accepted"] for r in trace] == [r["accepted"] for r in trb]
    assert np.abs(x - xr).max() <= 1e-7 and np.abs(x - xb).max() <= 1e-7
    prob.close()
    band.close()


def align_to_pose0(x, ref):
    """x with the rigid transform applied that maps its pose 0 onto ref's pose 0 (poses [N][12]: R row-major, then p): the BALM
    cost does not see a common rigid motion of all poses, an LM run without a gauge fix may drift along it."""
    x, ref = np.asarray(x, float).reshape(-1, 12), np.asarray(ref, float).reshape(-1, 12)
    R0, p0 = x[0, :9].reshape(3, 3), x[0, 9:]
    Rr, pr = ref[0, :9].reshape(3, 3), ref[0, 9:]
    T = Rr @ R0.T                       # T R0 = Rr
    t = pr - T @ p0                     # T p0 + t = pr
    out = np.empty_like(x)
    out[:, :9] = (T @ x[:, :9].reshape(-1, 3, 3)).reshape(-1, 9)
    out[:, 9:] = x[:, 9:] @ T.T + t
    return out.reshape(np.shape(ref))


@pytest.mark.parametrize("world,case", [(4, dict(n_poses=640, n_voxels=24000, band=6, seed=11)),
                                        (3, dict(n_poses=600, n_voxels=20000, band=6, seed=12, loop_frac=0.0)),
                                        # config C4's shape scaled down (n / bw > 20) on the eight ranks BASELINE.json gives it
                                        (8, dict(n_poses=1600, n_voxels=48000, band=5, seed=13))])
def test_long_band_shared_by_ranks(pkg, oracle_mod, world, case, monkeypatch):
    """Chunks of the band ordering as separators: every rank factorises its own arcs, the separator system is summed over the
    ranks and solved, every rank substitutes back into its arcs, the solution is summed.  Ranks bitwise equal; equal to the
    single-rank band solve to rounding; LM run equal to the oracle's."""
    d = make_problem(**case)
    N, off, idx, clu = d["n_poses"], d["voxel_off"], d["pose_idx"], d["clusters"]
    V = len(off) - 1
    monkeypatch.setenv("LVBA_SOLVER", "nond")
    single = pkg.BalmProblem(N, off, idx, clu)
    H1, g1, c1 = single.eval(d["poses_init"])
    dx1 = single.solve(0.01)
    x1, tr1, rc1 = single.refine(d["poses_init"])
    single.close()
    monkeypatch.setenv("LVBA_SOLVER", "nd")
    ht = HostTransport(world)

    def rank_main(r):
        a, b = pkg.shard_range(V, r, world)
        prob = pkg.BalmProblem(N, off[a:b + 1], idx[off[a]:off[b]], clu[off[a]:off[b]])
        ht.attach(prob, r)
        info = prob.info()
        H, g, c = prob.eval(d["poses_init"])
        dx = prob.solve(0.01)
        x, trace, rc = prob.refine(d["poses_init"])
        prob.close()
        return dict(info=info, H=H, g=g, c=c, dx=dx, x=x, trace=trace, rc=rc)

    out = ht.run(rank_main)
    r0 = out[0]
    assert r0["info"]["nd_kind"] == 2 and r0["info"]["nd_arcs"] >= world, r0["info"]
    if world == 8:
        assert N / max(1, r0["info"]["nd_sep_band_blocks"]) > 8
    for o in out[1:]:
        assert np.array_equal(o["H"], r0["H"]) and np.array_equal(o["dx"], r0["dx"]) and np.array_equal(o["x"], r0["x"])
        assert o["trace"] == r0["trace"]
    assert rel(r0["H"], H1) <= 1e-12 and rel(r0["g"], g1) <= 1e-12
    want = dense_solve(r0["H"], r0["g"], 0.01)
    assert np.abs(r0["dx"] - want).max() <= 1e-8 * np.abs(want).max()
    assert np.abs(r0["dx"] - dx1).max() <= 1e-8 * np.abs(want).max()
    assert r0["rc"] == rc1 == 0 and len(r0["trace"]) == len(tr1)
    # LM costs (gauge-invariant) and poses.  Worlds 3 / 4: 1e-7 throughout, against the single-rank band path and the oracle.
    # World 8 is a chain of 1 600 poses with next to no loop closures: ITS LM run amplifies rounding differences ~10 x per iteration
    # between ANY two implementations -- measured (round 6), relative cost difference per iteration, this path against the single-rank
    # band path / the single-rank path against the C oracle:  iterations 0 - 6: <= 5e-10 / <= 3e-8;  7: 2e-9 / 7e-7;  8: 3e-8 / 7e-6;
    # 9 (a REJECTED trial point): 5e-6 / 1e-3.  So against the single-rank path every accepted step's cost is held at 1e-7 and a
    # rejected trial point's at 1e-4; against the oracle the first seven iterations at 1e-7 and the converged cost at 1e-5.  The poses
    # are compared twice: after aligning both runs to their pose 0 (1e-6 for world 8: pose 0's own rounding in the
    # rotation, ~1e-9 rad, times the 200-m extent of the trajectory is what the alignment itself adds), and raw (north_star's 1e-5).
    big = world == 8
    tol_raw, tol_aligned = (1e-5, 1e-6) if big else (1e-7, 1e-7)
    assert np.abs(r0["x"] - x1).max() <= tol_raw, np.abs(r0["x"] - x1).max()
    assert np.abs(align_to_pose0(r0["x"], x1) - x1).max() <= tol_aligned, np.abs(align_to_pose0(r0["x"], x1) - x1).max()
    for a, b in zip(r0["trace"], tr1):
        assert a["accepted"] == b["accepted"]
        assert abs(a["residual1"] - b["residual1"]) <= 1e-7 * b["residual1"]
        assert abs(a["residual2"] - b["residual2"]) <= (1e-7 if a["accepted"] or not big else 1e-4) * b["residual2"]
    co = oracle_mod.COracle(N, off, idx, clu)
    xr, tr, _ = co.damping_iter(d["poses_init"])
    assert len(tr) == len(r0["trace"])
    for k, (a, b) in enumerate(zip(r0["trace"], tr)):   # (oracle trace row: iter, residual1, residual2, u, v, q, q1, accepted, evaluated)
        assert a["accepted"] == int(b[7])
        if not big or k < 7:
            assert abs(a["residual1"] - b[1]) <= 1e-7 * b[1] and abs(a["residual2"] - b[2]) <= 1e-7 * b[2]
    best = min(r["residual2"] if r["accepted"] else r["residual1"] for r in r0["trace"])
    best_ref = min(r[2] if r[7] else r[1] for r in tr)
    assert abs(best - best_ref) <= (1e-5 if big else 1e-7) * best_ref
    if not big:   # (world 8: the chain's gauge drifts 1e-4 between the single-rank band path and the oracle themselves)
        assert np.abs(r0["x"] - xr).max() <= tol_raw, np.abs(r0["x"] - xr).max()
        assert np.abs(align_to_pose0(r0["x"], xr) - xr).max() <= tol_aligned, np.abs(align_to_pose0(r0["x"], xr) - xr).max()
