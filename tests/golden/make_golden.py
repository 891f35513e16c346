"""Generates tests/golden/*.npz: small packed problems + the oracle's outputs on them.

The reference has no tests / golden vectors of its own.  Two kinds of fixtures are written here:
  * balm_*.npz, visual_small.npz, voxel_small.npz, tracks_small.npz -- inputs + the answers of OUR restatement
    (oracle/*.py): they freeze the oracle, any later change to oracle/ or to the generator that alters results is caught;
  * ref_balm.npz, ref_voxel.npz (main_ref), ref_system.npz (main_ref_system) -- the answers of THE REFERENCE'S OWN CODE on
    the same inputs: include/BALM/{tools,bavoxel}.hpp, and src/lvba_system.cpp + src/dataset_io.cpp (the whole pipeline up
    to ceres::Solve), compiled from /root/reference against the stand-ins of oracle/shim (oracle/_ref/*.so, `make -C oracle
    ref`).  These are what the -m gpu tests on the GPU box (where /root/reference does not exist) hold the HIP path
    against.  The iterations of the visual stage's solver live in Ceres (absent): visual_small.npz stays restatement-only.

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


def main_voxel():
    """Voxel front-end / window stage / track triangulation: inputs + what the oracles answer (frozen)."""
    synth = importlib.import_module("global-lvba_amd.synth")
    from oracle import voxel_oracle as vx, window_oracle as wo, track_oracle as to
    s = synth.make_scans(6, 2500, room=(8, 6, 3), origin=(-2.5, 4.0, 0.3), n_panels=6, seed=17, rot_sigma_deg=0.1, trans_sigma=0.03)
    flat = np.concatenate(s["clouds"]).astype(np.float32)
    counts = np.array([len(c) for c in s["clouds"]], np.int64)
    surf_map, vox = vx.build(s["clouds"], s["poses"], 1.0)
    off, idx, clu = vx.pack(vox)
    key = np.array([list(k) + [len(p) | ((p[0] if len(p) >= 1 else 0) << 4) | ((p[1] if len(p) == 2 else 0) << 8)]
                    for k, p, _ in vox], np.int64)
    rng = np.random.default_rng(1)
    Xq = np.concatenate([c[rng.choice(len(c), 40, replace=False), :3].astype(np.float64) @ T[:9].reshape(3, 3).T + T[9:]
                         for c, T in zip(s["clouds"], s["poses"])])
    planes = np.zeros((len(Xq), 4)); valid = np.zeros(len(Xq), np.uint8)
    for i, x in enumerate(Xq):
        r = vx.find_plane(surf_map, x, 1.0)
        if r is not None:
            planes[i, :3], planes[i, 3], valid[i] = r[0], r[1], 1
    w = wo.run_window_ba(s["clouds"], s["poses"], 3, 1.0, np.float32([0.3, 0.1, 0.06, 0.03]), 0.05)
    np.savez_compressed(os.path.join(HERE, "voxel_small.npz"), points=flat, counts=counts, poses=s["poses"], voxel_off=off,
                        pose_idx=idx, clusters=clu, voxel_key=key, query=Xq, planes=planes, valid=valid,
                        win_anchor_index=w["anchor_index"], win_rel_poses=w["rel_poses"], win_window_poses=w["window_poses"],
                        win_anchor_poses=w["anchor_poses"], win_anchor_counts=np.array([len(c) for c in w["anchor_clouds"]]))
    print("voxel_small", len(flat), "points ->", len(off) - 1, "voxels,", int(valid.sum()), "of", len(Xq), "queries on planes,",
          len(w["anchor_clouds"]), "anchors")
    d = synth.make_visual_problem(8, 60, seed=3, track_len=5)
    q = d["q_gt"]
    qw, qx, qy, qz = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    Rcw = np.stack([1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy),
                    2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx),
                    2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)], 1).reshape(-1, 3, 3)
    ok, X, err, cnt = to.triangulate_tracks(d["intr"], Rcw, d["t_gt"], d["obs_off"], d["obs_cam"], d["obs_uv"])
    np.savez_compressed(os.path.join(HERE, "tracks_small.npz"), Rcw=Rcw, tcw=d["t_gt"], obs_off=d["obs_off"], obs_cam=d["obs_cam"],
                        obs_uv=d["obs_uv"], intr=d["intr"], ok=ok, X=X, mean_reproj=err, count=cnt)
    print("tracks_small", int(ok.sum()), "of", len(ok), "tracks triangulated")


def main_ref():
    """Reference-generated golden vectors: needs /root/reference (or a prebuilt oracle/_ref/libbalm_ref.so)."""
    import oracle
    ref = oracle.Reference()
    out = {}
    for name in CASES:
        z = np.load(os.path.join(HERE, name + ".npz"))
        slots = oracle.csr_to_slots(int(z["n_poses"]), z["voxel_off"], z["pose_idx"], z["clusters"])
        H, g, c_avg = ref.divide_thread(slots, z["poses_init"])             # BALM2::divide_thread
        out[name + "__H"], out[name + "__g"], out[name + "__cost_avg"] = H, g, c_avg
        out[name + "__cost_sum"] = ref.only_residual(slots, z["poses_init"], False)
        xf = ref.damping_iter(slots, z["poses_init"])                       # BALM2::damping_iter
        out[name + "__poses_final"] = xf
        out[name + "__cost_final_avg"] = ref.only_residual(slots, xf, True)
        print("ref", name, "cost", c_avg, "->", out[name + "__cost_final_avg"])
    np.savez_compressed(os.path.join(HERE, "ref_balm.npz"), **out)
    z = np.load(os.path.join(HERE, "voxel_small.npz"))
    clouds = np.split(z["points"], np.cumsum(z["counts"])[:-1])
    m = ref.map_build(clouds, z["poses"], 1.0)                              # cut_voxel + recut + tras_opt
    adm = np.array([np.count_nonzero(c[:, 9] != 0) >= 2 for c in m["clusters"]])
    planes = ref.map_find_planes(m["handle"], z["query"], 1.0)              # findCorrespondPoint at the call site
    ref.map_free(m["handle"])
    # anchor cloud of the first window of 3 frames merged with the (odometry) poses, then down_sampling_voxel2
    np.savez_compressed(os.path.join(HERE, "ref_voxel.npz"), keys=m["keys"][adm], slots=m["clusters"][adm], geo=m["geo"][adm],
                        n_roots=m["n_roots"], n_plane_nodes=len(adm), n_admitted=m["n_admitted"], planes=planes)
    print("ref voxel_small", m["n_roots"], "roots,", len(adm), "plane nodes,", int(adm.sum()), "admitted,",
          int(np.any(planes != 0, axis=1).sum()), "queries on planes")


def main_fusion():
    """LiDAR-assisted landmark initialisation: scans + camera rig + tracks and what oracle/fusion_oracle.py answers (frozen)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_fusion as T
    from oracle import fusion_oracle as fo
    clouds, poses, times, img_t, Rcw, tcw = T._scene(n_frames=4, pts=14000, n_cams=6)
    rng = np.random.default_rng(12)
    off, img, uv, X = T._tracks(clouds, poses, Rcw, tcw, rng, n_tracks=90)
    depth = fo.render_depth(clouds, poses, times, img_t, Rcw, tcw, T.INTR, T.W, T.H, half_w=100.0)
    depth_win = fo.render_depth(clouds, poses, times, img_t, Rcw, tcw, T.INTR, T.W, T.H)       # the reference's +-0.5 s windows
    st, Xf, err, kept = fo.fuse_tracks(off, img, uv, depth, Rcw, tcw, T.INTR)
    np.savez_compressed(os.path.join(HERE, "fusion_small.npz"), points=np.concatenate(clouds).astype(np.float32)[:, :3],
                        counts=np.array([len(c) for c in clouds], np.int64), scan_poses=poses, scan_times=times, image_times=img_t,
                        Rcw=Rcw, tcw=tcw, intr=T.INTR, width=T.W, height=T.H, obs_off=off, obs_img=img, obs_uv=uv,
                        depth_filled=(depth > 0).sum(axis=(1, 2)), depth_sum=depth.astype(np.float64).sum(axis=(1, 2)),
                        depth_win_filled=(depth_win > 0).sum(axis=(1, 2)), status=st, X=Xf, mean_reproj=err, kept=kept)
    print("fusion_small", np.bincount(st, minlength=3).tolist(), "dropped / triangulated / depth-fused")


def main_ref_system():
    """The whole pipeline of the reference (src/lvba_system.cpp + src/dataset_io.cpp compiled against oracle/shim, see
    oracle/ref_glue_system.cpp) on the synthetic sequence of tests/test_gpu_pipeline.py: what ITS stages answer, frozen for the
    GPU test (needs /root/reference)."""
    import importlib
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_pipeline as tp
    import test_ref_system as trs
    from oracle import ref_system as rs
    ds = importlib.import_module("global-lvba_amd.dataset")
    d = tp._dataset()
    root = os.path.join(tempfile.mkdtemp(), "seq")
    trs.write_sequence(root, d, ds)
    trs.INTR, trs.W, trs.H = tp.INTR, tp.W, tp.H
    S = rs.ReferenceSystem(root, trs.reference_params(tp))
    L = ds.load_dataset(root)
    _, img_poses = ds.load_poses_tum(os.path.join(root, "all_image", "image_poses.txt"), 1)
    S.init()
    S.run_lidar_ba()
    R1, p1, ts = S.scan_poses()
    S.build_grid_map(); S.update_camera_poses()
    depth = S.generate_depth(tp.W, tp.H)
    Rcw, tcw = S.cam_poses(True)
    S.set_features(d["kps"], {pr: m for pr, m in zip(d["pairs"], d["matches"])})
    tracks = S.build_tracks()
    P = S.optimize()
    np.savez_compressed(os.path.join(HERE, "ref_system.npz"), cloud_digest=np.array([float(np.asarray(c, np.float64).sum()) for c in d["clouds"]]),
                        scan_poses_in=L["poses"], image_poses_in=img_poses, scan_times=ts, image_times=S.image_ids(),
                        scan_poses_out=np.concatenate([R1.reshape(-1, 9), p1], 1), Rcw=Rcw, tcw=tcw,
                        depth_filled=(depth > 0).sum(axis=(1, 2)), depth_sum=depth.astype(np.float64).sum(axis=(1, 2)),
                        track_start=np.array([t["obs"][0] for t in tracks], np.int32), track_len=np.array([len(t["obs"]) for t in tracks]),
                        track_X=np.array([t["X"] for t in tracks]), track_inliers=np.array([len(t["inliers"]) for t in tracks]),
                        n_points=P["n_points"], n_residuals=len(P["kind"]), cost0=P["cost0"])
    print("ref_system", len(tracks), "tracks,", P["n_points"], "with a plane, cost0", P["cost0"])
    S.close()
    shutil.rmtree(os.path.dirname(root), ignore_errors=True)


if __name__ == "__main__":
    main()
    main_visual()
    main_voxel()
    main_ref()
    main_fusion()
    main_ref_system()
