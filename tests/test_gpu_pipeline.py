"""End-to-end GPU test of the host-side pipeline mirror (global-lvba_amd/pipeline.py = LvbaSystem::runFullPipeline,
src/lvba_system.cpp:136-154) on a synthetic dataset in memory: noisy odometry + raw scans + keypoints / matches in, refined
LiDAR poses, cameras and landmarks out -- every compute step through the C-ABI on the GPU."""
import importlib

import numpy as np
import pytest

from oracle import track_oracle as to

pytestmark = pytest.mark.gpu

RCB = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0]])      # camera z = body y: looking sideways, so that
# the motion along the trajectory gives the tracks parallax (the view-angle filter wants >= 8 degrees between kept rays)
TCI = np.array([0.02, 0.05, -0.03])
INTR = np.array([300.0, 298.0, 240.0, 180.0, -0.076160, 0.123001, -0.00113, 0.000251])   # the reference camera's distortion (a
# negative k2 would fold rays far outside the field of view back into the image and pollute the z-buffer)
W, H = 480, 360


def _dataset(n_frames=24, pts=80000, n_land=900, seed=61, INTR=INTR, W=W, H=H):
    synth = importlib.import_module("global-lvba_amd.synth")
    pipe = importlib.import_module("global-lvba_amd.pipeline")
    s = synth.make_scans(n_frames, pts, room=(14, 10, 4), n_panels=0, n_blobs=0, clutter_frac=0.0, seed=seed, rot_sigma_deg=0.15,
                         trans_sigma=0.04)
    gt = np.asarray(s["poses_gt"], np.float64).reshape(-1, 12)
    odo = np.asarray(s["poses"], np.float64).reshape(-1, 12)
    times = 50.0 + 0.1 * np.arange(n_frames)
    img_t = times + 0.004
    rng = np.random.default_rng(seed)
    world = np.concatenate([c[:, :3].astype(np.float64) @ T[:9].reshape(3, 3).T + T[9:] for c, T in zip(s["clouds"], gt)])
    X = world[rng.choice(len(world), n_land, replace=False)]
    Rcw_gt, tcw_gt = pipe.camera_from_imu(gt, RCB, TCI)
    kps, lm_of = [], []
    for m in range(n_frames):
        k, ids = [], []
        for li, x in enumerate(X):
            p = to.project(INTR, Rcw_gt[m], tcw_gt[m], x)
            if p is not None and 3 < p[0] < W - 4 and 3 < p[1] < H - 4 and (Rcw_gt[m] @ x + tcw_gt[m])[2] < 12.0:
                k.append(np.float32(p) + np.float32(0.4 * rng.standard_normal(2))); ids.append(li)
        kps.append(np.array(k, np.float32).reshape(-1, 2)); lm_of.append(ids)
    pairs, matches = [], []
    for i in range(n_frames):
        for j in range(i + 1, min(n_frames, i + 7)):
            pos_j = {li: kj for kj, li in enumerate(lm_of[j])}
            m = [(ki, pos_j[li]) for ki, li in enumerate(lm_of[i]) if li in pos_j]
            if m:
                pairs.append((i, j)); matches.append(np.array(m, np.int64))
    return dict(clouds=s["clouds"], gt=gt, odo=odo, times=times, img_t=img_t, X=X, kps=kps, lm_of=lm_of, pairs=pairs,
                matches=matches, Rcw_gt=Rcw_gt, tcw_gt=tcw_gt)


def _rel_err(a, b):
    """Largest error of the positions relative to frame 0 (gauge-free)."""
    pa = (a[:, 9:] - a[0, 9:]) @ a[0, :9].reshape(3, 3)
    pb = (b[:, 9:] - b[0, 9:]) @ b[0, :9].reshape(3, 3)
    return np.abs(pa - pb).max()


def test_full_pipeline_on_a_synthetic_dataset(pkg):
    pipe = importlib.import_module("global-lvba_amd.pipeline")
    d = _dataset()
    out = pipe.run_full_pipeline(d["clouds"], d["odo"], d["times"], d["img_t"], d["odo"], RCB, TCI, INTR, W, H, d["kps"],
                                 d["pairs"], d["matches"], window_size=6, anchor_leaf=0.02, stage_voxel_size=(1.0, 0.5),
                                 stage_eigen_ratio=((0.2,) * 4, (0.08,) * 4))
    # LiDAR stage: the refined trajectory is closer to the ground truth than the odometry
    rep = out["lidar_report"]
    assert rep["n_windows"] == 4 and rep["n_anchors"] >= 3 and rep["stage_ran"][1] == 1
    e0, e1 = _rel_err(d["odo"], d["gt"]), _rel_err(out["poses"], d["gt"])
    assert e1 < 0.5 * e0, (e0, e1)
    v = out["visual"]
    # tracks: a good part of the components is fused, by both candidates
    st = v["track_status"]
    assert v["n_components"] > 300 and (st > 0).mean() > 0.5 and (st == 1).sum() > 20 and (st == 2).sum() > 20
    # landmarks land on the scene and most of them find a plane of the refined map
    assert v["landmark_valid"].mean() > 0.6
    # every fused landmark is close to some true scene point (the landmark set it was generated from)
    from scipy.spatial import cKDTree
    dist, _ = cKDTree(d["X"]).query(v["landmarks"][v["landmark_valid"] > 0])
    assert np.median(dist) < 0.08
    # visual stage: the solve converged and reduced its cost; the cameras stay consistent with the refined LiDAR poses
    tr = v["trace"]
    assert v["termination"].startswith("CONVERGENCE") and tr[-1]["cost"] < 0.7 * tr[0]["cost"]
    Rcw_l, tcw_l = v["Rcw_lidar"], v["tcw_lidar"]
    Cw_l = -np.einsum("nji,nj->ni", Rcw_l, tcw_l)
    Cw_v = -np.einsum("nji,nj->ni", v["Rcw"], v["tcw"])
    assert np.abs(Cw_v - Cw_l).max() < 0.1
    # and both are closer to the true camera centres than the odometry was (relative to camera 0)
    Cw_gt = -np.einsum("nji,nj->ni", d["Rcw_gt"], d["tcw_gt"])
    Cw_0 = -np.einsum("nji,nj->ni", v["Rcw_before"], v["tcw_before"])
    rel = lambda C: C - C[0]
    assert np.abs(rel(Cw_v) - rel(Cw_gt)).max() < np.abs(rel(Cw_0) - rel(Cw_gt)).max()


def test_pipeline_from_a_dataset_directory(pkg, tmp_path):
    """The same flow through the reference's on-disk formats (src/dataset_io.cpp, loadFromColmapDB): TUM pose files, binary
    PCDs named by time stamp, image files named by time stamp (only the names are read), a COLMAP database with keypoints and
    two_view_geometries.  The result must agree with the in-memory run up to the precision of the text pose files."""
    import sqlite3
    pipe = importlib.import_module("global-lvba_amd.pipeline")
    ds = importlib.import_module("global-lvba_amd.dataset")
    d = _dataset(n_frames=12, pts=40000, n_land=400, seed=62)
    root = tmp_path / "seq"
    (root / "all_pcd_body").mkdir(parents=True); (root / "all_image").mkdir()
    for t, c in zip(d["times"], d["clouds"]):
        ds.save_pcd(str(root / "all_pcd_body" / f"{t:.6f}.pcd"), np.concatenate([c[:, :3], np.zeros((len(c), 1), np.float32)], 1))
    ds.write_poses_tum(str(root / "all_pcd_body" / "lidar_poses.txt"), d["times"], d["odo"])
    for t in d["img_t"]:
        (root / "all_image" / f"{t:.6f}.png").write_bytes(b"")
    ds.write_poses_tum(str(root / "all_image" / "image_poses.txt"), d["img_t"], d["odo"])
    con = sqlite3.connect(str(root / "colmap.db"))
    con.execute("CREATE TABLE images (image_id INTEGER PRIMARY KEY, name TEXT)")
    con.execute("CREATE TABLE keypoints (image_id INTEGER PRIMARY KEY, rows INTEGER, cols INTEGER, data BLOB)")
    con.execute("CREATE TABLE two_view_geometries (pair_id INTEGER PRIMARY KEY, rows INTEGER, cols INTEGER, data BLOB)")
    for i, t in enumerate(d["img_t"]):
        kp4 = np.concatenate([d["kps"][i], np.ones((len(d["kps"][i]), 2), np.float32)], 1).astype(np.float32)
        con.execute("INSERT INTO images VALUES (?, ?)", (i + 1, f"{t:.6f}.png"))
        con.execute("INSERT INTO keypoints VALUES (?, ?, ?, ?)", (i + 1, kp4.shape[0], 4, kp4.tobytes()))
    for (i, j), m in zip(d["pairs"], d["matches"]):
        con.execute("INSERT INTO two_view_geometries VALUES (?, ?, ?, ?)",
                    (ds.image_ids_to_pair_id(i + 1, j + 1), len(m), 2, np.asarray(m, np.uint32).tobytes()))
    con.commit(); con.close()
    cfg = dict(window_size=6, anchor_leaf=0.02, stage_voxel_size=(1.0, 0.5), stage_eigen_ratio=((0.2,) * 4, (0.08,) * 4))
    # Rcl / Pcl with an identity lidar->imu extrinsic give Rci = RCB, tci = TCI
    got = pipe.run_dataset(str(root), "colmap.db", INTR, W, H, RCB, TCI, out_dir=str(tmp_path / "out"), **cfg)
    ref = pipe.run_full_pipeline(d["clouds"], d["odo"], d["times"], d["img_t"], d["odo"], RCB, TCI, INTR, W, H, d["kps"], d["pairs"],
                                 d["matches"], **cfg)
    assert np.abs(got["poses"] - ref["poses"]).max() < 1e-5
    gv, rv = got["visual"], ref["visual"]
    assert gv["n_components"] == rv["n_components"] and np.array_equal(gv["track_status"], rv["track_status"])
    assert np.abs(gv["tcw"] - rv["tcw"]).max() < 1e-3
    out = tmp_path / "out"
    assert (out / "lidar_poses_refined.txt").exists() and (out / "images.txt").exists() and (out / "points3D.txt").exists()
    assert len((out / "images.txt").read_text().splitlines()) == 2 * len(d["img_t"])
    back_t, back = ds.load_poses_tum(str(out / "lidar_poses_refined.txt"), 1)
    assert np.abs(back - got["poses"]).max() < 1e-5
    Rci, tci = pipe.extrinsics_from_config(RCB, TCI, np.eye(3), np.zeros(3))
    assert np.array_equal(Rci, RCB) and np.array_equal(tci, TCI)


def test_pipeline_matches_the_reference_system_golden(pkg):
    """tests/golden/ref_system.npz holds what the reference's OWN pipeline (src/lvba_system.cpp + src/dataset_io.cpp compiled
    against the stand-ins of oracle/shim, tests/golden/make_golden.py:main_ref_system) answers on the synthetic sequence of
    _dataset(): refined scan poses, cameras, depth image digests, the fused tracks in its order and the cost of the problem it
    hands to ceres::Solve.  The HIP pipeline gets the same inputs (the poses as the reference parsed them from its TUM files)."""
    import os
    pipe = importlib.import_module("global-lvba_amd.pipeline")
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_system.npz"))
    d = _dataset()
    assert np.array_equal(np.array([float(np.asarray(c, np.float64).sum()) for c in d["clouds"]]), z["cloud_digest"])  # same scans
    out = pipe.run_full_pipeline(d["clouds"], z["scan_poses_in"], z["scan_times"], z["image_times"], z["image_poses_in"], RCB, TCI, INTR,
                                 W, H, d["kps"], d["pairs"], d["matches"], window_size=6, anchor_leaf=0.02, use_rel=True,
                                 stage_voxel_size=(1.0, 0.5), stage_eigen_ratio=((0.2,) * 4, (0.08,) * 4))
    # LiDAR stage (runLidarBA): window BA + two global stages, two LM loops deep
    assert np.abs(out["poses"] - z["scan_poses_out"]).max() < 1e-5
    v = out["visual"]
    assert np.abs(v["Rcw_lidar"] - z["Rcw"]).max() < 1e-5 and np.abs(v["tcw_lidar"] - z["tcw"]).max() < 1e-5
    # tracks (BuildTracksAndFuse3D): index work -- the bar is exact.  The depth renderer, the fusion and the triangulation are
    # compiled without FMA contraction (build.py: NO_CONTRACT), like the plain x86-64 build the golden file came from, so every
    # depth pixel, every threshold decision and hence every track agrees: the same tracks in the same order.
    T = v["tracks"]
    starts = np.stack([T["obs_img"][T["obs_off"][:-1]], T["obs_kp"][T["obs_off"][:-1]]], 1)
    assert len(T["X"]) == len(z["track_X"]), (len(T["X"]), len(z["track_X"]))
    assert np.array_equal(starts, z["track_start"])
    assert np.abs(T["X"] - z["track_X"]).max() < 1e-5, float(np.abs(T["X"] - z["track_X"]).max())   # the scan poses agree to 1e-5 only
    assert np.array_equal(np.diff(T["obs_off"]), z["track_len"])
    inl = np.add.reduceat(T["kept"].astype(np.int64), T["obs_off"][:-1])
    assert np.array_equal(inl, z["track_inliers"])
    assert (T["attempts"] > 0).sum() >= 10                                   # the reference's retries are exercised
    # the problem handed to the solver: landmarks with a plane, and the cost at the initial point (it is a smooth function of
    # the refined scan poses, which agree to 1e-5 m over two LM loops; whitened residuals are O(1/0.01 m) per metre)
    assert int(v["landmark_valid"].sum()) == int(z["n_points"])
    assert abs(v["trace"][0]["cost"] - float(z["cost0"])) <= 1e-3 * float(z["cost0"]), (v["trace"][0]["cost"], float(z["cost0"]))
