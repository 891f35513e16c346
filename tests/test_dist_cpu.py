"""world_size-2 gloo test (CPU): the factor sharding of SURVEY.md 8(e).  Each rank evaluates its
contiguous voxel range with the oracle, the pose-block H / g / cost are all-reduced, and the result must
equal the unsharded evaluation -- the invariant the RCCL path in liblvba_hip.so relies on
(bavoxel.hpp:621-633 with thread -> rank)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, make_problem


def _worker(rank, world, port, d, out):
    sys.path.insert(0, ROOT)
    import importlib
    import oracle
    pkg = importlib.import_module("global-lvba_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off = d["voxel_off"]
    V = len(off) - 1
    a, b = pkg.shard_range(V, rank, world)
    co = oracle.COracle(d["n_poses"], off[a:b + 1] - off[a], d["pose_idx"][off[a]:off[b]], d["clusters"][off[a]:off[b]])
    H, g, c = co.eval_dense(d["poses_init"])
    n = H.shape[0]
    buf = torch.from_numpy(np.concatenate([H.reshape(-1), g, [c * (b - a)], [float(b - a)]]))
    dist.all_reduce(buf)                                   # sum over ranks == the thread sum of bavoxel.hpp:626-633
    if rank == 0:
        out["H"] = buf[:n * n].reshape(n, n).numpy().copy()
        out["g"] = buf[n * n:n * n + n].numpy().copy()
        out["cost_avg"] = float(buf[-2] / buf[-1])
        out["V"] = float(buf[-1])
    dist.destroy_process_group()


def test_sharded_eval_allreduce_equals_whole():
    import oracle
    oracle.build_c()
    d = make_problem(12, 61, band=4, seed=31)            # odd voxel count: uneven shards
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(2, port, d, out), nprocs=2, join=True)
    co = oracle.COracle(d["n_poses"], d["voxel_off"], d["pose_idx"], d["clusters"])
    H, g, c = co.eval_dense(d["poses_init"])
    assert out["V"] == 61
    assert np.abs(out["H"] - H).max() <= 1e-12 * np.abs(H).max()
    assert np.abs(out["g"] - g).max() <= 1e-12 * np.abs(g).max()
    assert abs(out["cost_avg"] - c) <= 1e-12 * c
