"""Which pose block of the Hessian is furthest from the C oracle, and why?  (C4's worst block sits at 5.4e-8 while
C2 / C3 hold 1e-8.)  Evaluates a configuration on the GPU and with oracle/balm_oracle.c, names the worst block, and looks at the
voxels the two poses share: the eigenvalue gaps of their merged covariances (the weight 2 / (lam0 - lam_m) of the in-plane terms
amplifies rounding where lam1 - lam0 is small) and how many factors are summed into the block.
usage: python tools/worst_block.py [C3|C4|NxV]   -> one JSON line"""
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import oracle  # noqa: E402

pkg = importlib.import_module("global-lvba_amd")
synth = importlib.import_module("global-lvba_amd.synth")


def merged_eigs(d, x0, voxels):
    """eigenvalues of the merged world-frame covariance of each voxel (tools.hpp:450-456, bavoxel.hpp:97-98), numpy / LAPACK"""
    off, idx, clu = d["voxel_off"], d["pose_idx"], d["clusters"]
    out = []
    for v in voxels:
        S = np.zeros((3, 3)); m = np.zeros(3); n = 0.0
        for f in range(off[v], off[v + 1]):
            R = x0[idx[f], :9].reshape(3, 3); p = x0[idx[f], 9:]
            c = clu[f]
            P = np.array([[c[0], c[1], c[2]], [c[1], c[3], c[4]], [c[2], c[4], c[5]]]); vv = c[6:9]; k = c[9]
            Rv = R @ vv
            S += R @ P @ R.T + np.outer(Rv, p) + np.outer(p, Rv) + k * np.outer(p, p)
            m += Rv + k * p
            n += k
        C = S / n - np.outer(m / n, m / n)
        out.append(np.linalg.eigvalsh(C))
    return np.array(out)


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
    N, V = bench.parse_config(cfg, synth)
    d = synth.make_balm_problem(N, V, device="cuda:0")
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    prob = pkg.BalmProblem(N, d["voxel_off"], d["pose_idx"], d["clusters"], device=0)
    x0 = d["poses_init"]
    gi, gj, gb, g, c = prob.eval_blocks(x0)
    co = oracle.COracle(N, d["voxel_off"], d["pose_idx"], d["clusters"])
    bi, bj, blocks, gc, cc = co.eval_sparse(x0, nthreads=min(16, os.cpu_count() or 1))
    worst, outside, det = oracle.block_parity_sparse(gi, gj, gb, bi, bj, blocks, N, details=True)
    i, j = det["pose_i"], det["pose_j"]
    off, idx = d["voxel_off"], d["pose_idx"]
    vox_of_f = np.repeat(np.arange(V), np.diff(off))
    vi, vj = set(vox_of_f[idx == i].tolist()), set(vox_of_f[idx == j].tolist())
    shared = sorted(vi & vj)
    ev = merged_eigs(d, x0, shared[:2000])
    gap = (ev[:, 1] - ev[:, 0]) / ev[:, 2]
    # the same statistics over a random sample of voxels, for comparison
    rng = np.random.default_rng(0)
    ev_all = merged_eigs(d, x0, rng.choice(V, 2000, replace=False).tolist())
    gap_all = (ev_all[:, 1] - ev_all[:, 0]) / ev_all[:, 2]
    out = {"config": cfg, "H_block_rel_worst": worst, "worst_block": det, "shared_voxels": len(shared),
           "worst_block_gap_min": float(gap.min()), "worst_block_gap_median": float(np.median(gap)),
           "sample_gap_min": float(gap_all.min()), "sample_gap_median": float(np.median(gap_all)),
           "worst_block_lam0_over_lam1_max": float((ev[:, 0] / ev[:, 1]).max()),
           "note": "gap = (lam1 - lam0) / lam2 of a voxel's merged covariance; the in-plane weights 2 / (lam0 - lam_m) amplify the "
                   "rounding of lam0 and of u0 where the gap is small"}
    print(json.dumps(out), flush=True)
    prob.close()


if __name__ == "__main__":
    main()
