"""GPU parity AT THE BASELINE.json CONFIGURATION SIZES: the HIP path (through the C-ABI) against the C oracle
(oracle/balm_oracle.c, itself pinned against the reference's own bavoxel.hpp, tests/test_ref_pin.py) on the very problems
bench.py times.

  C2 (500 poses x 400k voxels x 2M factors)   in full: cost, gradient, EVERY non-zero pose block of the Hessian
                                              (bavoxel.hpp:68-174), and the whole damping_iter trace (:662-767)
  C3 (2000 poses x 2M voxels x 10M factors)   one 400k-voxel shard in full (cost, g, every block), the cost of the full
                                              problem, and the whole damping_iter trace of the full problem

At these sizes the chunk table, voxels split over workgroups, the pair-list splitting and the band / dense switch of the
solver all take other branches than in the 150-pose cases of test_gpu_balm.py.  Tolerances: 1e-8 on cost / g / H blocks
(fp64 everywhere; lambda_min is a 1e8:1 cancellation), 1e-7 on the per-iteration LM costs, 1e-5 on poses (north_star).
The oracle's dense LM is out of reach here (12 000^2), so the trace comes from its sparse / band twin
(bo_damping_iter_band), which test_oracle.py holds bit-equal to the dense one on small problems.
"""
import os

import numpy as np
import pytest

from test_gpu_balm import _compare_traces

pytestmark = pytest.mark.gpu


def _gen(synth, name):
    import torch
    N, V = synth.CONFIGS[name]
    d = synth.make_balm_problem(N, V, device="cuda")
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return N, V, d


def _threads():
    return min(16, os.cpu_count() or 1)


def test_c2_full_parity(pkg, synth, oracle_mod):
    N, V, d = _gen(synth, "C2")
    prob = pkg.BalmProblem(N, d["voxel_off"], d["pose_idx"], d["clusters"])
    co = oracle_mod.COracle(N, d["voxel_off"], d["pose_idx"], d["clusters"])
    for x in (d["poses_init"], d["poses_gt"]):
        c_ref = co.cost(x, nthreads=_threads())
        assert abs(prob.cost(x) - c_ref) <= 1e-8 * c_ref
    x = d["poses_init"]
    H, g, c = prob.eval(x)
    bi, bj, blocks, gc, cc = co.eval_sparse(x, nthreads=_threads())
    assert abs(c - cc) <= 1e-8 * cc
    assert np.abs(g - gc).max() <= 1e-8 * np.abs(gc).max()
    worst, outside = oracle_mod.block_parity(H, bi, bj, blocks)
    assert worst <= 1e-8, worst
    assert outside <= 1e-12, outside
    assert np.array_equal(H, H.T)
    del H
    # the whole LM loop, row by row
    xg, trace, rc = prob.refine(x)
    xr, tr, rcr, _ = co.damping_iter_band(x, perm=prob.ordering(), eval_threads=_threads())
    assert rc == 0 and rcr == 0
    _compare_traces(trace, tr, xg, xr)
    prob.close()


def test_c3_shard_cost_and_lm_trace(pkg, synth, oracle_mod):
    N, V, d = _gen(synth, "C3")
    off, idx, clu = d["voxel_off"], d["pose_idx"], d["clusters"]
    x = d["poses_init"]
    # (1) one 400k-voxel shard (the slicing formula of bavoxel.hpp:621-624: slice 2 of 5), in full
    a, b = pkg.shard_range(V, 2, 5)
    assert b - a == 400_000
    sh = pkg.BalmProblem(N, off[a:b + 1], idx[off[a]:off[b]], clu[off[a]:off[b]])
    cs = oracle_mod.COracle(N, off[a:b + 1] - off[a], idx[off[a]:off[b]], clu[off[a]:off[b]])
    H, g, c = sh.eval(x)
    bi, bj, blocks, gc, cc = cs.eval_sparse(x, nthreads=_threads())
    assert abs(c - cc) <= 1e-8 * cc
    assert np.abs(g - gc).max() <= 1e-8 * np.abs(gc).max()
    worst, outside = oracle_mod.block_parity(H, bi, bj, blocks)
    assert worst <= 1e-8, worst
    assert outside <= 1e-12, outside
    del H
    sh.close()
    # (2) the full problem: cost at two points, then the whole damping_iter trace
    prob = pkg.BalmProblem(N, off, idx, clu)
    co = oracle_mod.COracle(N, off, idx, clu)
    for xx in (x, d["poses_gt"]):
        c_ref = co.cost(xx, nthreads=_threads())
        assert abs(prob.cost(xx) - c_ref) <= 1e-8 * c_ref
    xg, trace, rc = prob.refine(x)
    xr, tr, rcr, sec = co.damping_iter_band(x, perm=prob.ordering(), eval_threads=_threads())
    assert rc == 0 and rcr == 0
    assert prob.info()["use_band"] == 1                       # the band path of the solver is what ran
    _compare_traces(trace, tr, xg, xr)
    prob.close()


def test_c4_cost_and_one_lm_iteration(pkg, synth, oracle_mod):
    """C4 (10 000 poses x 10 M voxels x 50 M factors -- the configuration BASELINE.json assigns to 8 GPUs) on ONE GPU: cost at
    two points and the gradient of the full problem, every pose block of one 400k-voxel shard (slice 12 of 25 by the formula of
    bavoxel.hpp:621-624), every pose block of the FULL problem through the sparse export (the dense Hessian would be 28.8 GB),
    and the first LM iteration -- evaluation, damped band solve (n = 60 000), retraction, cost at the trial point -- against
    the oracle's band twin."""
    N, V, d = _gen(synth, "C4")
    off, idx, clu = d["voxel_off"], d["pose_idx"], d["clusters"]
    x = d["poses_init"]
    a, b = pkg.shard_range(V, 12, 25)
    assert b - a == 400_000
    sh = pkg.BalmProblem(N, off[a:b + 1], idx[off[a]:off[b]], clu[off[a]:off[b]])
    cs = oracle_mod.COracle(N, off[a:b + 1] - off[a], idx[off[a]:off[b]], clu[off[a]:off[b]])
    gi, gj, gblocks, g, c = sh.eval_blocks(x)
    bi, bj, blocks, gc, cc = cs.eval_sparse(x, nthreads=_threads())
    assert abs(c - cc) <= 1e-8 * cc
    assert np.abs(g - gc).max() <= 1e-8 * np.abs(gc).max()
    worst, extra = oracle_mod.block_parity_sparse(gi, gj, gblocks, bi, bj, blocks, N)
    assert worst <= 1e-8 and extra <= 1e-12, (worst, extra)
    del gblocks, blocks
    sh.close()
    prob = pkg.BalmProblem(N, off, idx, clu)
    co = oracle_mod.COracle(N, off, idx, clu)
    c_ref = co.cost(d["poses_gt"], nthreads=_threads())
    assert abs(prob.cost(d["poses_gt"]) - c_ref) <= 1e-8 * c_ref
    gi, gj, gblocks, g, c = prob.eval_blocks(x)
    bi, bj, blocks, gc, cc = co.eval_sparse(x, nthreads=_threads())
    assert abs(c - cc) <= 1e-8 * cc and abs(prob.cost(x) - cc * V) <= 1e-8 * cc * V    # (eval: averaged, cost: the sum)
    assert np.abs(g - gc).max() <= 1e-8 * np.abs(gc).max()
    worst, extra = oracle_mod.block_parity_sparse(gi, gj, gblocks, bi, bj, blocks, N)
    # 1.87 M blocks: the worst of them sits at 5e-8 (lambda_min is a 1e8 : 1 cancellation and a block sums ~60 voxels; the
    # shard above and C2 / C3 stay below 1.2e-8) -- the bar here is the one bench.py's parity gate uses, north_star asks 1e-5
    assert worst <= 1e-7 and extra <= 1e-12, (worst, extra)
    assert len(gi) == len(bi)
    del gblocks, blocks
    info = prob.info()
    assert info["use_band"] == 1 and info["n_factors"] == int(off[-1])
    xg, trace, rc = prob.refine(x, max_iter=1)
    xr, tr, rcr, sec = co.damping_iter_band(x, perm=prob.ordering(), max_iter=1, eval_threads=_threads(), solve_threads=min(32, os.cpu_count() or 1))
    assert rc == 0 and rcr == 0 and len(trace) == 1 and len(tr) == 1
    _compare_traces(trace, tr, xg, xr)
    prob.close()
