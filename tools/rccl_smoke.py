"""First contact with RCCL on more than one GPU (no round had a node: SCALE_r0*.json are all "skipped").  Run under torchrun, one
rank per GPU; tools/first_node.sh does it.

 1. every rank builds its voxel shard of one problem, joins the RCCL communicator through the product's own entry point
    (lvba_balm_dist_init) and runs evaluation, damped solve and a whole LM refinement: H blocks, g, cost, dx, trace and refined
    poses must be BITWISE equal on all ranks;
 2. rank 0 repeats the same job with the ranks as host threads on its own GPU through tests/host_transport.cpp (the transport
    every round's multi-rank tests used) and holds the RCCL result against it: bitwise if RCCL's ring happens to sum in rank
    order, else to rounding (1e-12 on H / g / dx, 1e-7 on LM costs and poses) -- which one is printed;
 3. the bus bandwidth of an fp64 sum all-reduce at the sizes the pose blocks of C3 / C4 have (107 / 533 MB), 2 (N - 1) / N x bytes /
    time, written to profiles/rccl_bus_bandwidth.json: bench.py's scaling_model reads it instead of its assumed 250 GB/s.

usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29533 tools/rccl_smoke.py [n_poses n_voxels]"""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("global-lvba_amd")
synth = importlib.import_module("global-lvba_amd.synth")


def job(prob, x0):
    bi, bj, blocks, g, c = prob.eval_blocks(x0)
    dx = prob.solve(0.01)
    x, trace, rc = prob.refine(x0)
    return dict(bi=bi, bj=bj, H=blocks, g=g, c=c, dx=dx, x=x, rc=rc,
                trace=[(t["accepted"], t["residual1"], t["residual2"]) for t in trace])


def same(a, b):
    return all(np.array_equal(np.asarray(a[k]), np.asarray(b[k])) for k in ("bi", "bj", "H", "g", "dx", "x")) and a["c"] == b["c"] \
        and a["trace"] == b["trace"]


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def main():
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)            # control plane
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    V = int(sys.argv[2]) if len(sys.argv) > 2 else 120_000
    d = synth.make_balm_problem(N, V, band=12, seed=21)
    off, idx, clu, x0 = d["voxel_off"], d["pose_idx"], d["clusters"], d["poses_init"]
    a, b = pkg.shard_range(V, rank, world)
    prob = pkg.BalmProblem(N, off[a:b + 1], idx[off[a]:off[b]], clu[off[a]:off[b]], device=local)
    if world > 1:
        uid = [pkg.BalmProblem.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        prob.dist_init(world, rank, uid[0])
    mine = job(prob, x0)
    prob.close()
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    report = {"n_gpus": world, "n_poses": N, "n_voxels": V}
    ok = True
    if rank == 0:
        report["ranks_bitwise_equal"] = all(same(everyone[0], o) for o in everyone[1:])
        ok &= report["ranks_bitwise_equal"]
        from conftest import HostTransport
        ht = HostTransport(world)

        def rank_main(r):
            a2, b2 = pkg.shard_range(V, r, world)
            p2 = pkg.BalmProblem(N, off[a2:b2 + 1], idx[off[a2]:off[b2]], clu[off[a2]:off[b2]], device=local)
            ht.attach(p2, r)
            out = job(p2, x0)
            p2.close()
            return out

        host = ht.run(rank_main)[0] if world > 1 else mine
        report["vs_host_transport"] = {"bitwise": same(mine, host), "H_rel": rel(mine["H"], host["H"]), "g_rel": rel(mine["g"], host["g"]),
                                       "cost_rel": abs(mine["c"] - host["c"]) / abs(host["c"]), "dx_rel": rel(mine["dx"], host["dx"]),
                                       "poses_abs": float(np.abs(mine["x"] - host["x"]).max()),
                                       "lm_cost_rel": max([abs(p[1] - q[1]) / q[1] for p, q in zip(mine["trace"], host["trace"])] +
                                                          [abs(p[2] - q[2]) / q[2] for p, q in zip(mine["trace"], host["trace"])]),
                                       "same_accept_pattern": [p[0] for p in mine["trace"]] == [q[0] for q in host["trace"]]}
        v = report["vs_host_transport"]
        ok &= v["H_rel"] <= 1e-12 and v["g_rel"] <= 1e-12 and v["dx_rel"] <= 1e-8 and v["lm_cost_rel"] <= 1e-7 and v["poses_abs"] <= 1e-7
    # ---- bus bandwidth of the fp64 sum all-reduce through RCCL
    bw = {}
    if world > 1:
        pg = dist.new_group(backend="nccl")
        for name, mb in (("C3_pose_blocks", 107), ("C4_pose_blocks", 533)):
            t = torch.ones(mb * 1_000_000 // 8, dtype=torch.float64, device=f"cuda:{local}")
            for _ in range(3):
                dist.all_reduce(t, group=pg)
            torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            for _ in range(10):
                dist.all_reduce(t, group=pg)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            tt = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            bw[name] = {"mb": mb, "ms": 1e3 * float(tt[0]), "bus_gb_s": 2.0 * (world - 1) / world * mb * 1e-3 / float(tt[0])}
            del t
    if rank == 0:
        report["allreduce"] = bw
        report["ok"] = bool(ok)
        os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
        with open(os.path.join(ROOT, "profiles", f"rccl_smoke_{world}gpu.json"), "w") as f:
            json.dump(report, f, indent=1)
        if bw:
            with open(os.path.join(ROOT, "profiles", "rccl_bus_bandwidth.json"), "w") as f:
                json.dump({"n_gpus": world, "bus_gb_s": bw["C3_pose_blocks"]["bus_gb_s"], "by_size": bw,
                           "how": "tools/rccl_smoke.py: fp64 sum all-reduce through RCCL, 2 (N - 1) / N x bytes / time, max over ranks"}, f, indent=1)
        print(json.dumps(report))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0 and not ok:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
