"""The damped solve on pose graphs that are NOT the synthetic ring-with-antipodal-chords of the BASELINE configs (VERDICT round 4,
item 5): synth's revisit = figure8 / chords / lot at N = 2 000 and N = 10 000 poses.  Per graph: the band ordering's width, which
solver path ran by default (band / dense / nested dissection) and how long a solve takes, the same with the dissection switched
off (LVBA_SOLVER=nond), the relative residual of the damped system  || (H + u diag H) dx + g || / || g ||  computed on the HOST
from the sparse pose blocks (lvba_balm_eval_blocks) for both, the two solutions against each other, and -- at N = 2 000 -- the
refined poses of a whole LM run of either path against each other.

    python tools/graph_bench.py [--sizes 2000,10000] [--graphs antipodal,figure8,chords,lot] [--voxels-per-pose 200]
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def residual(prob, x0, u, dx, N):
    import scipy.sparse as sp
    bi, bj, blocks, g, c = prob.eval_blocks(x0)
    n = 6 * N
    rows = (6 * bi[:, None, None] + np.arange(6)[None, :, None]).repeat(6, axis=2)
    cols = (6 * bj[:, None, None] + np.arange(6)[None, None, :]).repeat(6, axis=1)
    L = sp.coo_matrix((blocks.reshape(-1), (rows.reshape(-1), cols.reshape(-1))), shape=(n, n)).tocsr()
    off = bi != bj
    rows_t, cols_t = cols[off], rows[off]
    U = sp.coo_matrix((blocks[off].reshape(-1), (rows_t.reshape(-1), cols_t.reshape(-1))), shape=(n, n)).tocsr()
    H = L + U
    d = H.diagonal()
    r = H @ dx + u * d * dx + g
    return float(np.linalg.norm(r) / np.linalg.norm(g))


def run(pkg, d, N, mode, u=0.01, reps=5):
    import torch
    old = os.environ.get("LVBA_SOLVER")
    if mode == "nond":
        os.environ["LVBA_SOLVER"] = "nond"
    else:
        os.environ.pop("LVBA_SOLVER", None)
    try:
        t0 = time.perf_counter()
        prob = pkg.BalmProblem(N, d["voxel_off"], d["pose_idx"], d["clusters"])
        info = prob.info()
        setup_s = time.perf_counter() - t0
    finally:
        if old is None:
            os.environ.pop("LVBA_SOLVER", None)
        else:
            os.environ["LVBA_SOLVER"] = old
    x0 = d["poses_init"]
    prob.eval(x0, want_H=False, want_g=False)
    dx = prob.solve(u)
    prob.solve(u); prob.solve(u)
    prob.set_profiling(True); prob.profile(reset=True)
    for _ in range(reps):
        prob.solve(u)
    p = prob.profile()
    prob.set_profiling(False)
    out = {"path": ("nested dissection (%s: %d arcs, separator %d poses, separator band %d blocks)" %
                    ({1: "hubs", 2: "chunks"}[info["nd_kind"]], info["nd_arcs"], info["nd_sep_poses"], info["nd_sep_band_blocks"])
                    if info["nd_kind"] else ("band LDL^T" if info["use_band"] else "dense LDL^T")),
           "band_blocks_of_the_store": info["band_blocks"], "solve_ms": p["solve_ms"] / max(1, p["solve_calls"]),
           "setup_s": setup_s, "hess_store_gb": info["hess_bytes"] / 1e9,
           "model_ms": {"band": info["nd_model_band_ms"], "nd": info["nd_model_nd_ms"]} if info["nd_kind"] else None,
           "residual_rel": residual(prob, x0, u, dx.ravel(), N)}
    return prob, dx, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="2000,10000")
    ap.add_argument("--graphs", default="antipodal,figure8,chords,lot")
    ap.add_argument("--voxels-per-pose", type=int, default=200)
    args = ap.parse_args()
    pkg = importlib.import_module("global-lvba_amd")
    synth = importlib.import_module("global-lvba_amd.synth")
    res = []
    for N in [int(v) for v in args.sizes.split(",")]:
        for graph in args.graphs.split(","):
            d = synth.make_balm_problem(N, args.voxels_per_pose * N, revisit=graph, device="cuda")
            row = {"graph": graph, "n_poses": N, "n_factors": int(d["voxel_off"][-1])}
            pa, dxa, row["default"] = run(pkg, d, N, "auto")
            try:
                row["band_model"] = pa.nd_model(1)
            except Exception as e:
                row["band_model"] = repr(e)
            if row["default"]["path"].startswith("nested"):
                pb, dxb, row["without_dissection"] = run(pkg, d, N, "nond", reps=2 if N > 4000 else 5)
                row["dx_rel_between_paths"] = float(np.abs(dxa - dxb).max() / np.abs(dxb).max())
                if N <= 4000:
                    xa, ta, rca = pa.refine(d["poses_init"])
                    xb, tb, rcb = pb.refine(d["poses_init"])
                    row["lm"] = {"iterations": [len(ta), len(tb)], "rc": [rca, rcb], "poses_abs_between_paths": float(np.abs(xa - xb).max()),
                                 "accept_pattern_equal": [r["accepted"] for r in ta] == [r["accepted"] for r in tb]}
                pb.close()
            pa.close()
            print(json.dumps(row), flush=True)
            res.append(row)
    print(json.dumps({"graphs": res}))


if __name__ == "__main__":
    main()
