"""The product's MULTI-RANK code with N = 2 and 3 ranks on a one-GPU box.

RCCL refuses two ranks on one device, so the ranks are host threads of this process, each with its own handle (its own
voxel shard, its own stream) on the same GPU, joined through the library's external-transport entry point
(lvba_balm_dist_init_external) by a host-staged all-reduce that lives with the tests (tests/host_transport.cpp: device ->
host -> sum in rank order -> device).  Everything above the
transport is the code a real multi-GPU job runs (csrc/block_system.hip): the global voxel count, the max-reduced band
width, the all-reduced co-visibility matrix and the common pose order computed from it, the packed [H | g | cost]
all-reduce over the union sparsity pattern, the all-reduced cost scalar of the trial poses, the replicated damped solve.
Required: every rank ends with bitwise the same answer, and that answer equals the single-rank one to rounding
(partial sums are grouped differently) and the C oracle to 1e-8.  bavoxel.hpp:614-633 with thread -> rank."""
import numpy as np
import pytest

from conftest import HostTransport, make_problem, rel

pytestmark = pytest.mark.gpu


@pytest.fixture
def synth():
    import importlib
    return importlib.import_module("global-lvba_amd.synth")


def _run_ranks(pkg, d, world, packed=True):
    N, off, idx, clu = d["n_poses"], d["voxel_off"], d["pose_idx"], d["clusters"]
    V = len(off) - 1
    ht = HostTransport(world)

    def rank_main(r):
        a, b = pkg.shard_range(V, r, world)
        prob = pkg.BalmProblem(N, off[a:b + 1], idx[off[a]:off[b]], clu[off[a]:off[b]])
        ht.attach(prob, r)
        info = prob.info()
        H, g, c = prob.eval(d["poses_init"])
        c_gt = prob.cost(d["poses_gt"])
        x, trace, rc = prob.refine(d["poses_init"])
        res = dict(info=info, H=H, g=g, c=c, c_gt=c_gt, x=x, trace=trace, rc=rc, perm=prob.ordering())
        prob.close()
        return res

    return ht.run(rank_main)


@pytest.mark.parametrize("world,case", [(2, dict(n_poses=200, n_voxels=6000, band=10, seed=7)),     # packed all-reduce, band solver
                                        (3, dict(n_poses=40, n_voxels=3000, band=10, seed=2)),      # dense store, ragged shards
                                        (2, dict(n_poses=150, n_voxels=8000, band=12, seed=4)),
                                        # no loop closures -> a narrow band: ranks 0 and 1 eliminate one end of it each
                                        (2, dict(n_poses=200, n_voxels=6000, band=10, seed=7, loop_frac=0.0)),
                                        (3, dict(n_poses=300, n_voxels=9000, band=8, seed=9, loop_frac=0.0))])
def test_ranks_agree_with_single_rank_and_oracle(pkg, oracle_mod, world, case, monkeypatch):
    d = make_problem(**case)
    N = d["n_poses"]
    single = pkg.BalmProblem(N, d["voxel_off"], d["pose_idx"], d["clusters"])
    H1, g1, c1 = single.eval(d["poses_init"])
    cgt1 = single.cost(d["poses_gt"])
    x1, tr1, rc1 = single.refine(d["poses_init"])
    out = _run_ranks(pkg, d, world)
    r0 = out[0]
    assert r0["info"]["n_ranks"] == world and r0["info"]["n_voxels_global"] == len(d["voxel_off"]) - 1
    assert sum(o["info"]["n_voxels"] for o in out) == len(d["voxel_off"]) - 1
    if N >= 200 and "loop_frac" not in case:   # band + sparse far blocks: only the union pattern's blocks travel
        assert 0 < r0["info"]["allreduce_bytes"] < 0.75 * r0["info"]["hess_bytes"]     # the packed form was used
    if case.get("loop_frac") == 0.0:   # band systems: ranks 0 and 1 eliminate one end each, the middle block is exchanged (csrc/ldlt.hip)
        assert r0["info"]["use_band"] == 1 and r0["info"]["twist_panels"] >= 4 and r0["info"]["solve_ranks"] == 2
    for o in out[1:]:                                                                    # replicas: bitwise
        assert np.array_equal(o["perm"], r0["perm"])
        assert np.array_equal(o["H"], r0["H"]) and np.array_equal(o["g"], r0["g"]) and o["c"] == r0["c"] and o["c_gt"] == r0["c_gt"]
        assert np.array_equal(o["x"], r0["x"]) and o["trace"] == r0["trace"]
    # against the single-rank run: same sums, grouped per shard
    assert rel(r0["H"], H1) <= 1e-12 and rel(r0["g"], g1) <= 1e-12 and abs(r0["c"] - c1) <= 1e-12 * c1
    assert abs(r0["c_gt"] - cgt1) <= 1e-12 * cgt1
    assert r0["rc"] == rc1 == 0 and len(r0["trace"]) == len(tr1)
    for a, b in zip(r0["trace"], tr1):
        assert a["accepted"] == b["accepted"] and a["evaluated"] == b["evaluated"]
        # (rounding differences of the first evaluation grow from iteration to iteration of an LM run)
        assert abs(a["residual1"] - b["residual1"]) <= 1e-7 * b["residual1"] and abs(a["residual2"] - b["residual2"]) <= 1e-7 * b["residual2"]
    assert np.abs(r0["x"] - x1).max() <= 1e-8
    # and the oracle
    co = oracle_mod.COracle(N, d["voxel_off"], d["pose_idx"], d["clusters"])
    Hc, gc, cc = co.eval_dense(d["poses_init"])
    assert rel(r0["H"], Hc) <= 1e-8 and rel(r0["g"], gc) <= 1e-8 and abs(r0["c"] - cc) <= 1e-8 * cc
    xr, tr, _ = co.damping_iter(d["poses_init"])
    assert np.abs(r0["x"] - xr).max() <= 1e-7
    single.close()


def test_rank_with_a_different_band(pkg):
    """Shards whose LOCAL co-visibility differs (one rank holds only narrow-band voxels, the other the loop closures): the
    store layout, the pose order and the packed slot table must come from the GLOBAL pattern on both ranks."""
    d = make_problem(200, 6000, band=10, seed=7, loop_frac=0.3)
    off, idx = d["voxel_off"], d["pose_idx"]
    span = np.maximum.reduceat(idx, off[:-1]) - np.minimum.reduceat(idx, off[:-1])
    order = np.argsort(span, kind="stable")                       # narrow voxels first -> rank 0 sees no loop closure
    k = np.diff(off)
    new_off = np.concatenate([[0], np.cumsum(k[order])])
    gather = np.concatenate([np.arange(off[v], off[v + 1]) for v in order])
    dd = dict(d, voxel_off=new_off, pose_idx=idx[gather], clusters=d["clusters"][gather])
    single = pkg.BalmProblem(d["n_poses"], dd["voxel_off"], dd["pose_idx"], dd["clusters"])
    H1, g1, c1 = single.eval(d["poses_init"])
    out = _run_ranks(pkg, dd, 2)
    assert out[0]["info"]["band_blocks"] == out[1]["info"]["band_blocks"]
    assert np.array_equal(out[0]["H"], out[1]["H"])
    assert rel(out[0]["H"], H1) <= 1e-12 and rel(out[0]["g"], g1) <= 1e-12
    single.close()


@pytest.mark.parametrize("world,n_cams,n_tracks", [(2, 40, 1500), (3, 12, 300)])
def test_visual_track_shards_agree_with_single_rank(pkg, synth, world, n_cams, n_tracks):
    """The visual stage with its landmark tracks sharded over ranks (cameras replicated; lvba_visual_dist_init): the reduced
    camera system, the per-camera sums behind the LM diagonal and the Jacobi scaling, and the scalars of the trust-region loop are
    all-reduced; every rank must take the same decisions and return bitwise the same cameras, equal to the single-rank solve."""
    d = synth.make_visual_problem(n_cams, n_tracks, seed=11)
    off, cam, uv = d["obs_off"], d["obs_cam"], d["obs_uv"]
    one = pkg.VisualProblem(n_cams, off, cam, uv, d["plane"], d["valid"], d["intr"])
    c1 = one.cost(d["q"], d["t"], d["X"])
    S1, rhs1, _ = one.linearize(d["q"], d["t"], d["X"], radius=3.0)
    (q1, t1, X1), tr1, term1, rc1 = one.refine(d["q"], d["t"], d["X"])
    one.close()
    ht = HostTransport(world)

    def rank_main(r):
        a, b = pkg.shard_range(n_tracks, r, world)
        vp = pkg.VisualProblem(n_cams, off[a:b + 1], cam[off[a]:off[b]], uv[off[a]:off[b]], d["plane"][a:b], d["valid"][a:b], d["intr"])
        ht.attach(vp, r)
        c = vp.cost(d["q"], d["t"], d["X"][a:b])
        S, rhs, _ = vp.linearize(d["q"], d["t"], d["X"][a:b], radius=3.0)
        (q, t, X), tr, term, rc = vp.refine(d["q"], d["t"], d["X"][a:b])
        res = dict(c=c, S=S, rhs=rhs, q=q, t=t, X=X, tr=tr, term=term, rc=rc, a=a, b=b)
        vp.close()
        return res

    out = ht.run(rank_main)
    r0 = out[0]
    for o in out[1:]:
        assert o["c"] == r0["c"] and np.array_equal(o["S"], r0["S"]) and np.array_equal(o["rhs"], r0["rhs"])
        assert np.array_equal(o["q"], r0["q"]) and np.array_equal(o["t"], r0["t"]) and o["term"] == r0["term"]
        assert [row["cost"] for row in o["tr"]] == [row["cost"] for row in r0["tr"]]
    assert abs(r0["c"] - c1) <= 1e-12 * c1
    assert rel(r0["S"], S1) <= 1e-11 and rel(r0["rhs"], rhs1) <= 1e-11
    assert r0["rc"] == rc1 == 0 and r0["term"] == term1 and len(r0["tr"]) == len(tr1)
    for a_, b_ in zip(r0["tr"], tr1):
        assert a_["accepted"] == b_["accepted"] and abs(a_["cost"] - b_["cost"]) <= 1e-8 * b_["cost"]
    assert np.abs(r0["q"] - q1).max() <= 1e-8 and np.abs(r0["t"] - t1).max() <= 1e-8
    X = np.concatenate([o["X"] for o in out])
    assert np.abs(X - X1).max() <= 1e-7


def test_c3_two_ranks_full_lm_trace(pkg, synth):
    """The multi-rank path AT THE SIZE THE DRIVER LAUNCHES IT: config C3 (2 000 poses x 2 M voxels x 10 M factors) on two
    ranks.  What only happens at this size: the packed all-reduce over ~4e5 union-pattern blocks (115 MB), the two-rank split
    of the band factorisation with its 55 MB exchange of the middle block, windowed pair lists per shard, no solve graph.
    Required: both ranks bitwise equal; H blocks, g, cost and the whole LM trace equal to the single-rank run (1e-7 on the
    per-iteration costs, the bar of tests/test_gpu_config_parity.py, which holds the single-rank C3 run against the oracle)."""
    import torch
    N, V = synth.CONFIGS["C3"]
    d = synth.make_balm_problem(N, V, device="cuda")
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    off, idx, clu = d["voxel_off"], d["pose_idx"], d["clusters"]
    x0 = d["poses_init"]
    single = pkg.BalmProblem(N, off, idx, clu)
    gi1, gj1, B1, g1, c1 = single.eval_blocks(x0)
    x1, tr1, rc1 = single.refine(x0)
    single.close()
    world = 2
    ht = HostTransport(world)

    def rank_main(r):
        a, b = pkg.shard_range(V, r, world)
        prob = pkg.BalmProblem(N, off[a:b + 1], idx[off[a]:off[b]], clu[off[a]:off[b]])
        ht.attach(prob, r)
        info = prob.info()
        gi, gj, B, g, c = prob.eval_blocks(x0)
        x, trace, rc = prob.refine(x0)
        prob.close()
        return dict(info=info, gi=gi, gj=gj, B=B, g=g, c=c, x=x, trace=trace, rc=rc)

    out = ht.run(rank_main, timeout=900)
    r0, r1 = out
    calls, nbytes = ht.stats()
    assert r0["info"]["n_ranks"] == 2 and r0["info"]["n_voxels_global"] == V and r0["info"]["solve_ranks"] == 2
    assert 50e6 < r0["info"]["allreduce_bytes"] < 0.75 * r0["info"]["hess_bytes"]       # the packed form, at size
    assert nbytes > 5 * r0["info"]["allreduce_bytes"]                                    # ... and it did travel, every evaluation
    assert np.array_equal(r0["B"], r1["B"]) and np.array_equal(r0["g"], r1["g"]) and r0["c"] == r1["c"]
    assert np.array_equal(r0["x"], r1["x"]) and r0["trace"] == r1["trace"]
    # against the single-rank run: the union pattern may carry blocks that are zero on both shards' sum? no: the same blocks
    k1 = gi1.astype(np.int64) * N + gj1
    k0 = r0["gi"].astype(np.int64) * N + r0["gj"]
    o1, o0 = np.argsort(k1), np.argsort(k0)
    assert np.array_equal(k1[o1], k0[o0])
    assert rel(r0["B"][o0], B1[o1]) <= 1e-11 and rel(r0["g"], g1) <= 1e-11 and abs(r0["c"] - c1) <= 1e-12 * c1
    assert r0["rc"] == rc1 == 0 and len(r0["trace"]) == len(tr1)
    for a_, b_ in zip(r0["trace"], tr1):
        assert a_["accepted"] == b_["accepted"] and a_["evaluated"] == b_["evaluated"]
        assert abs(a_["residual1"] - b_["residual1"]) <= 1e-7 * b_["residual1"] and abs(a_["residual2"] - b_["residual2"]) <= 1e-7 * b_["residual2"]
    assert np.abs(r0["x"] - x1).max() <= 1e-7
