"""Generates tests/golden/*.npz: small packed problems + the oracle's outputs on them.

PARITY UNPINNED: the reference has no tests / golden vectors and cannot be built or imported here
(C++ needing Eigen, PCL, Ceres), so these fixtures come from OUR restatement (oracle/balm_oracle.py,
cross-checked against oracle/balm_oracle.c and finite differences), not from the reference itself.
They freeze the oracle: any later change to oracle/ or to the generator that alters results is caught.

    python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "balm_small": dict(n_poses=12, n_voxels=60, band=4, seed=1),
    "balm_window": dict(n_poses=20, n_voxels=150, band=20, seed=5),
    "balm_reject": dict(n_poses=12, n_voxels=60, band=4, seed=1, rot_sigma_deg=0.03, trans_sigma=0.02),
}


def main():
    synth = importlib.import_module("global-lvba_amd.synth")
    from oracle import balm_oracle as bo
    for name, kw in CASES.items():
        d = synth.make_balm_problem(**kw)
        prob = bo.Problem(d["n_poses"], d["voxel_off"], d["pose_idx"], d["clusters"])
        x0 = d["poses_init"]
        H, g, c = bo.divide_thread(prob, x0)
        xf, trace = bo.damping_iter(prob, x0)
        tr = np.array([[r.it, r.residual1, r.residual2, r.u, r.v, r.q, r.q1, r.accepted, r.evaluated] for r in trace])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), n_poses=d["n_poses"], voxel_off=d["voxel_off"],
                            pose_idx=d["pose_idx"], clusters=d["clusters"], poses_init=x0, poses_gt=d["poses_gt"],
                            cost_sum=bo.only_residual(prob, x0), cost_avg=c, H=H, g=g, poses_final=xf, trace=tr)
        print(name, "F =", len(d["pose_idx"]), "iters =", len(trace), "final cost", tr[-1, 2])


def main_visual():
    synth = importlib.import_module("global-lvba_amd.synth")
    from oracle import visual_oracle as vo
    d = synth.make_visual_problem(8, 60, seed=3)
    o = vo.VisualOracle(vo.VisualProblem(d["q"], d["t"], d["X"], d["obs_off"], d["obs_cam"], d["obs_uv"], d["plane"],
                                         d["valid"], d["intr"]))
    (q, t, X), trace, status = o.solve()
    np.savez_compressed(os.path.join(HERE, "visual_small.npz"), q=d["q"], t=d["t"], X=d["X"], obs_off=d["obs_off"],
                        obs_cam=d["obs_cam"], obs_uv=d["obs_uv"], plane=d["plane"], valid=d["valid"], intr=d["intr"],
                        cost0=o.cost(*o.state()), trace_cost=np.array([r["cost"] for r in trace]),
                        trace_radius=np.array([r["radius"] for r in trace]), q_final=q, t_final=t, X_final=X)
    print("visual_small", status, len(trace), trace[-1]["cost"])


if __name__ == "__main__":
    main()
    main_visual()
