"""CPU tests of the host-side glue of global-lvba_amd/pipeline.py (no GPU): camera poses from the refined LiDAR poses
(updateCameraPosesFromLidar, src/lvba_system.cpp:412-446), extrinsics composition (:484-504, :860-861), the BFS components
of the match graph (:923-1003) and the image listing of DatasetIO::handleImages (src/dataset_io.cpp:77-131)."""
import importlib

import numpy as np


def test_update_camera_poses_from_lidar_and_components():
    pipe = importlib.import_module("global-lvba_amd.pipeline")
    rng = np.random.default_rng(2)
    n = 6
    def rnd():
        from oracle import balm_oracle as bo
        return np.concatenate([bo.exp_so3(0.2 * rng.standard_normal(3)).reshape(-1), rng.standard_normal(3)])
    x_orig = np.array([rnd() for _ in range(n)]); x_opt = np.array([rnd() for _ in range(n)])
    cams = np.array([rnd() for _ in range(4)])
    ts = np.arange(n) * 1.0
    img_t = np.array([-3.0, 1.4, 1.6, 9.0])                                # before the first, nearer 1, nearer 2, after the last
    got = pipe.update_camera_poses_from_lidar(x_opt, x_orig, ts, img_t, cams)
    for i, idx in enumerate([0, 1, 2, 5]):
        Ro, po = x_opt[idx, :9].reshape(3, 3), x_opt[idx, 9:]
        Rb, pb = x_orig[idx, :9].reshape(3, 3), x_orig[idx, 9:]
        Rc, pc = cams[i, :9].reshape(3, 3), cams[i, 9:]
        Rd = Ro @ Rb.T
        assert np.abs(got[i, :9].reshape(3, 3) - Rd @ Rc).max() < 1e-14
        assert np.abs(got[i, 9:] - (Rd @ pc + po - Rd @ pb)).max() < 1e-13
    # identical trajectories leave the cameras untouched
    same = pipe.update_camera_poses_from_lidar(x_orig, x_orig, ts, img_t, cams)
    assert np.abs(same - cams).max() < 1e-13
    # components keep duplicates of an image and drop what is too small
    off, img, kp = pipe.build_components([3, 3, 3], [(0, 1), (1, 2), (0, 2)],
                                         [np.array([[0, 0], [1, 1], [2, 0]]), np.array([[0, 0]]), np.array([[1, 2]])])
    comps = [list(zip(img[a:b].tolist(), kp[a:b].tolist())) for a, b in zip(off[:-1], off[1:])]
    assert comps == [[(0, 0), (1, 0), (0, 2), (2, 0)], [(0, 1), (1, 1), (2, 2)]]




def test_extrinsics_camera_from_imu_and_image_listing(tmp_path):
    pipe = importlib.import_module("global-lvba_amd.pipeline")
    rng = np.random.default_rng(4)
    from oracle import balm_oracle as bo
    Rcl, Ril = bo.exp_so3(rng.standard_normal(3)), bo.exp_so3(0.3 * rng.standard_normal(3))
    tcl, til = rng.standard_normal(3), rng.standard_normal(3)
    Rci, tci = pipe.extrinsics_from_config(Rcl, tcl, Ril, til)
    # a lidar-frame point p_l: imu p_i = Ril p_l + til; camera p_c = Rcl p_l + tcl must equal Rci p_i + tci
    p_l = rng.standard_normal(3)
    assert np.abs(Rci @ (Ril @ p_l + til) + tci - (Rcl @ p_l + tcl)).max() < 1e-13
    # camera_from_imu: a world point X seen from imu pose (R, p): X_i = R^T (X - p); X_c = Rci X_i + tci
    T = np.concatenate([bo.exp_so3(rng.standard_normal(3)).reshape(-1), rng.standard_normal(3)])[None]
    Rcw, tcw = pipe.camera_from_imu(T, Rci, tci)
    X = rng.standard_normal(3)
    R, p = T[0, :9].reshape(3, 3), T[0, 9:]
    assert np.abs(Rcw[0] @ X + tcw[0] - (Rci @ (R.T @ (X - p)) + tci)).max() < 1e-13
    # quaternion round trip used around the visual solve
    q = pipe.rot_to_quat_wxyz(Rcw)
    assert np.abs(pipe.quat_wxyz_to_rot(q) - Rcw).max() < 1e-14 and abs(np.linalg.norm(q[0]) - 1) < 1e-15
    # image listing: numeric ids of image files, sorted, every stride-th; other files ignored
    for name in ["3.500000.png", "1.250000.jpg", "2.000000.bmp", "notes.txt", "image_poses.txt", "0.750000.jpeg", "abc.png"]:
        (tmp_path / name).write_bytes(b"")
    assert pipe.list_image_ids(str(tmp_path)).tolist() == [0.75, 1.25, 2.0, 3.5]
    assert pipe.list_image_ids(str(tmp_path), 2).tolist() == [0.75, 2.0]


def test_track_loop_edge_cases():
    """build_tracks_and_fuse with nothing to do, with a fusion that always fails (every member of every component is tried
    once, in scan order) and with one that succeeds on the second attempt (the track starts at the second member)."""
    pipe = importlib.import_module("global-lvba_amd.pipeline")
    kps = [np.zeros((3, 2), np.float32) for _ in range(4)]
    T = pipe.build_tracks_and_fuse(kps, [], [], lambda o, i, u: (_ for _ in ()).throw(AssertionError("no batch expected")))
    assert len(T["X"]) == 0 and T["obs_off"].tolist() == [0] and len(T["component_status"]) == 0
    pairs = [(0, 1), (1, 2), (2, 3)]
    matches = [np.array([[0, 0]]), np.array([[0, 0]]), np.array([[0, 0]])]           # one chain (0,0)-(1,0)-(2,0)-(3,0)
    starts = []

    def never(off, img, uv):
        starts.append(int(img[0]))
        n = len(off) - 1
        return np.zeros(n, np.uint8), np.zeros((n, 3)), np.full(n, np.inf), np.zeros(len(img), np.uint8)
    T = pipe.build_tracks_and_fuse(kps, pairs, matches, never)
    assert starts == [0, 1, 2, 3] and len(T["X"]) == 0 and T["component_status"].tolist() == [0]

    def second(off, img, uv):
        n = len(off) - 1
        ok = np.array([1 if img[off[k]] == 1 else 0 for k in range(n)], np.uint8)
        return ok, np.ones((n, 3)), np.zeros(n), np.ones(len(img), np.uint8)
    T = pipe.build_tracks_and_fuse(kps, pairs, matches, second)
    assert T["attempts"].tolist() == [1] and T["obs_img"].tolist() == [1, 0, 2, 3] and T["component_status"].tolist() == [1]
