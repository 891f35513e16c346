"""Pins the restatements of the pipeline AROUND the two LM loops against the reference's own src/lvba_system.cpp and
src/dataset_io.cpp, compiled unmodified against the stand-ins of oracle/shim (oracle/ref_glue_system.cpp ->
oracle/_ref/liblvba_system_ref.so).  A small synthetic sequence is written in the reference's on-disk layout, the
reference's LvbaSystem loads it and runs its stages one at a time; after every stage the restatement that the GPU tests use as
their oracle (oracle/*.py), and the product's host mirror (global-lvba_amd/pipeline.py, dataset.py), are held against it:

    DatasetIO                                -> dataset.load_dataset / load_poses_tum / pipeline.list_image_ids
    initFromDatasetIO                        -> pipeline.extrinsics_from_config
    runLidarBA (runWindowBA + two stages)    -> oracle.window_oracle.run_lidar_ba                     (poses to 1e-9)
    updateCameraPosesFromLidar               -> pipeline.update_camera_poses_from_lidar, camera_from_imu
    buildGridMapFromOptimized + generateDepthWithVoxel -> oracle.fusion_oracle.render_depth            (bit-identical images)
    BuildTracksAndFuse3D                     -> oracle.fusion_oracle.build_tracks_and_fuse             (same tracks, same order)
    optimizeCameraPoses up to ceres::Solve   -> window_oracle.merge_anchors + voxel_oracle.find_plane + visual_oracle cost
                                                (the problem Ceres would receive: blocks, constancy, manifold, residuals)

    loadFromColmapDB                         -> dataset.load_colmap_db                                 (system libsqlite3 linked)
    VisualizeOptComparison (COLMAP text)     -> dataset.write_images_txt / write_points3d_txt

What stays unpinned: the iterations of ceres::Solve (no Ceres here); SIFT extraction / matching is out of scope."""
import importlib
import os
import sys
import types

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import fusion_oracle as fo, ref_system as rs, visual_oracle as vis, voxel_oracle as vo, window_oracle as wo

pytestmark = pytest.mark.skipif(not rs.available(), reason="no reference sources and no prebuilt oracle/_ref/liblvba_system_ref.so")

INTR = np.array([150.0, 149.0, 120.0, 90.0, -0.076160, 0.123001, -0.00113, 0.000251])   # a small image, so that a modest cloud
W, H = 240, 180                                                                          # gives dense depth images
CFG = dict(window_size=6, anchor_leaf=0.02, stage_voxel_size=(1.0, 0.5), stage_eigen_ratio=((0.2,) * 4, (0.08,) * 4))


def write_sequence(root, d, ds):
    """The reference's dataset layout (README "Dataset", src/dataset_io.cpp) for the synthetic sequence d."""
    os.makedirs(os.path.join(root, "all_pcd_body")); os.makedirs(os.path.join(root, "all_image"))
    for t, c in zip(d["times"], d["clouds"]):
        ds.save_pcd(os.path.join(root, "all_pcd_body", f"{t:.6f}.pcd"), np.concatenate([c[:, :3], np.zeros((len(c), 1), np.float32)], 1))
    ds.write_poses_tum(os.path.join(root, "all_pcd_body", "lidar_poses.txt"), d["times"], d["odo"])
    for t in d["img_t"]:
        open(os.path.join(root, "all_image", f"{t:.6f}.png"), "wb").close()
    ds.write_poses_tum(os.path.join(root, "all_image", "image_poses.txt"), d["img_t"], d["odo"])


def reference_params(tp, cfg=CFG):
    """The ROS parameters (config/*.yaml names of the reference) for the synthetic camera / extrinsics / BALM settings."""
    I = INTR
    p = {"data_config/image_sample_step": 1, "cam_model/cam_width": W, "cam_model/cam_height": H, "cam_model/scale": 1.0,
         "extrin_calib/extrinsic_T": [0, 0, 0], "extrin_calib/extrinsic_R": np.eye(3), "extrin_calib/Pcl": tp.TCI, "extrin_calib/Rcl": tp.RCB,
         "window_ba/enable": True, "window_ba/size": cfg["window_size"], "window_ba/anchor_leaf_size": cfg["anchor_leaf"],
         "window_ba/use_window_ba_rel": True, "BALM_stage1/root_voxel_size": cfg["stage_voxel_size"][0],
         "BALM_stage1/eigen_ratio_array": cfg["stage_eigen_ratio"][0], "BALM_stage2/root_voxel_size": cfg["stage_voxel_size"][1],
         "BALM_stage2/eigen_ratio_array": cfg["stage_eigen_ratio"][1]}
    for k, v in zip(("fx", "fy", "cx", "cy", "d0", "d1", "d2", "d3"), I):
        p["cam_model/cam_" + k] = float(v)
    return p


@pytest.fixture(scope="module")
def run(tmp_path_factory):
    import test_gpu_pipeline as tp
    ds = importlib.import_module("global-lvba_amd.dataset")
    pipe = importlib.import_module("global-lvba_amd.pipeline")
    d = tp._dataset(n_frames=12, pts=16000, n_land=260, seed=63, INTR=INTR, W=W, H=H)
    root = str(tmp_path_factory.mktemp("refsys") / "seq")
    write_sequence(root, d, ds)
    S = rs.ReferenceSystem(root, reference_params(tp))
    r = types.SimpleNamespace(tp=tp, ds=ds, pipe=pipe, d=d, root=root, S=S)
    r.loaded = ds.load_dataset(root)
    r.R0, r.p0, r.ts = S.scan_poses()
    r.clouds_ref = [S.cloud(i) for i in range(S.n_clouds)]
    r.image_ids = S.image_ids()
    r.img_R0, r.img_t0 = S.image_poses()
    r.camera = S.camera()
    r.Rci, r.tci = S.init()
    r.anchor_idx, r.rel_R, r.rel_p = S.run_lidar_ba()
    r.R1, r.p1, _ = S.scan_poses()
    r.x_opt = np.concatenate([r.R1.reshape(-1, 9), r.p1], 1)
    r.grid_points, r.n_voxel_ids = S.build_grid_map()
    r.cam_R, r.cam_t = S.update_camera_poses()
    r.depth = S.generate_depth(W, H)
    r.Rcw, r.tcw = S.cam_poses(True)
    r.Rcw_before, r.tcw_before = S.cam_poses(False)
    S.set_features(d["kps"], {pr: m for pr, m in zip(d["pairs"], d["matches"])})
    r.tracks = S.build_tracks()
    r.problem = S.optimize()
    yield r
    S.close()


def test_dataset_io(run):
    r = run
    assert r.S.n_scans == r.S.n_clouds == r.S.n_images == 12
    L = r.loaded
    assert np.abs(L["poses"][:, :9].reshape(-1, 3, 3) - r.R0).max() < 1e-15 and np.array_equal(L["poses"][:, 9:], r.p0)
    assert np.array_equal(L["timestamps"], r.ts)                                   # parseTimestampFromName of the file names
    for a, b in zip(L["clouds"], r.clouds_ref):
        assert np.array_equal(np.asarray(a)[:, :3], b[:, :3])
    assert np.array_equal(r.pipe.list_image_ids(os.path.join(r.root, "all_image")), r.image_ids)
    _, img = r.ds.load_poses_tum(os.path.join(r.root, "all_image", "image_poses.txt"), 1)
    assert np.abs(img[:, :9].reshape(-1, 3, 3) - r.img_R0).max() < 1e-15 and np.array_equal(img[:, 9:], r.img_t0)
    assert r.camera["width"] == W and r.camera["height"] == H and np.array_equal(r.camera["intr"], INTR)
    Rci, tci = r.pipe.extrinsics_from_config(r.tp.RCB, r.tp.TCI, np.eye(3), np.zeros(3))
    assert np.abs(Rci - r.Rci).max() < 1e-15 and np.abs(tci - r.tci).max() < 1e-15


def test_run_lidar_ba(run):
    """runLidarBA = runWindowBA + stage 1 + stage 2 + the anchor -> scan composition, against the restatement the GPU tests of
    lvba_lidar_ba / lvba_window_ba use."""
    r = run
    L = r.loaded
    out, rep = wo.run_lidar_ba(L["clouds"], L["poses"], window_size=CFG["window_size"], anchor_leaf=CFG["anchor_leaf"], use_rel=True,
                               stage_voxel_size=CFG["stage_voxel_size"], stage_eigen_ratio=CFG["stage_eigen_ratio"])
    assert np.abs(r.p1 - r.p0).max() > 0.02                                        # the reference did move the trajectory
    assert np.abs(out[:, :9].reshape(-1, 3, 3) - r.R1).max() < 1e-9 and np.abs(out[:, 9:] - r.p1).max() < 1e-9
    w = wo.run_window_ba(L["clouds"], L["poses"], CFG["window_size"], CFG["stage_voxel_size"][0], np.float32((0.3, 0.1, 0.06, 0.03)),
                         CFG["anchor_leaf"], True)
    assert np.array_equal(w["anchor_index"], r.anchor_idx)
    assert np.abs(w["rel_poses"][:, :9].reshape(-1, 3, 3) - r.rel_R).max() < 1e-9 and np.abs(w["rel_poses"][:, 9:] - r.rel_p).max() < 1e-9


def test_camera_poses_and_depth_images(run):
    r = run
    L = r.loaded
    _, img = r.ds.load_poses_tum(os.path.join(r.root, "all_image", "image_poses.txt"), 1)
    cam_new = r.pipe.update_camera_poses_from_lidar(r.x_opt, L["poses"], r.ts, r.image_ids, img)
    assert np.abs(cam_new[:, :9].reshape(-1, 3, 3) - r.cam_R).max() < 1e-13 and np.abs(cam_new[:, 9:] - r.cam_t).max() < 1e-13
    Rcw, tcw = r.pipe.camera_from_imu(cam_new, r.Rci, r.tci)
    assert np.abs(Rcw - r.Rcw).max() < 1e-13 and np.abs(tcw - r.tcw).max() < 1e-12
    Rb, tb = r.pipe.camera_from_imu(img, r.Rci, r.tci)
    assert np.abs(Rb - r.Rcw_before).max() < 1e-14 and np.abs(tb - r.tcw_before).max() < 1e-13
    assert r.grid_points == sum(len(c) for c in L["clouds"]) and min(r.n_voxel_ids) > 100
    depth = fo.render_depth(L["clouds"], r.x_opt, r.ts, r.image_ids, r.Rcw, r.tcw, INTR, W, H)
    assert (r.depth > 0).mean() > 0.1
    assert np.array_equal(depth, r.depth)                                          # every pixel of every image, bit for bit


def test_unordered_map_order_is_the_real_one(tmp_path):
    """fusion_oracle.umap_order / UMAP_BUCKETS against std::unordered_map<int,int> of this toolchain."""
    import ctypes
    import subprocess
    src = tmp_path / "umap.cpp"
    src.write_text('#include <cstddef>\n#include <unordered_map>\nextern "C" int real_order(int res, int m, const int *k, int *out) {\n'
                   "  std::unordered_map<int, int> u; u.reserve((std::size_t)res);\n  for (int i = 0; i < m; ++i) u[k[i]] = i;\n"
                   "  int n = 0; for (const auto &kv : u) out[n++] = kv.second; return (int)u.bucket_count(); }\n")
    so = str(tmp_path / "umap.so")
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-o", so, str(src)])
    lib = ctypes.CDLL(so)
    ip = ctypes.POINTER(ctypes.c_int)
    rng = np.random.default_rng(5)
    for _ in range(600):
        m = int(rng.integers(1, 48))
        res = m + int(rng.integers(0, 40))
        keys = rng.choice(int(rng.integers(m, 4000)), m, replace=False).astype(np.int32)
        out = np.zeros(m, np.int32)
        B = lib.real_order(res, m, keys.ctypes.data_as(ip), out.ctypes.data_as(ip))
        assert B == fo.umap_bucket_count(res) and fo.umap_order(keys.tolist(), res) == out.tolist()
    for res in (3, 200, 5000, 150000):
        out = np.zeros(1, np.int32)
        assert lib.real_order(res, 1, out.ctypes.data_as(ip), out.ctypes.data_as(ip)) == fo.umap_bucket_count(res)


def test_build_tracks_and_fuse(run):
    """Same tracks in the same order, the same observations in the same BFS order, the same inlier sets, landmarks to rounding --
    including the components the reference only fuses on a retry from a later member."""
    r = run
    mine = fo.build_tracks_and_fuse(r.d["kps"], r.d["pairs"], r.d["matches"], r.depth, r.Rcw, r.tcw, INTR)
    assert len(mine) == len(r.tracks) > 100
    n_tri = n_depth = 0
    for a, b in zip(r.tracks, mine):
        assert np.array_equal(a["obs"], b["obs"])
        assert np.abs(a["X"] - b["X"]).max() < 1e-10
        assert sorted(a["inliers"].tolist()) == np.nonzero(b["kept"])[0].tolist()
        n_tri += b["status"] == 1; n_depth += b["status"] == 2
    assert n_tri > 10 and n_depth > 10
    # the retry quirk is exercised: some track does not start at the smallest member of its component
    assert any(tuple(t["obs"][0]) != min(map(tuple, t["obs"].tolist())) for t in r.tracks)


def test_device_fusion_code_reproduces_the_reference_tracks(run, tmp_path_factory):
    """The per-track code the fuse kernel runs (global-lvba_amd/csrc/fusion_device.h, compiled for the host by
    tests/host_emul_tracks.cpp) on the components of the reference's tracks, in the reference's BFS order: same candidate
    selected, same kept observations, same landmark."""
    import test_tracks_host as th
    lib = th.build_emul(tmp_path_factory.mktemp("emul_tracks_ref"))
    r = run
    off = np.concatenate([[0], np.cumsum([len(t["obs"]) for t in r.tracks])]).astype(np.int64)
    obs = np.concatenate([t["obs"] for t in r.tracks])
    uv = np.array([r.d["kps"][i][k][:2] for i, k in obs], np.float32)
    st, X, err, kept = th._fuse(lib, off, obs[:, 0], uv, r.depth, r.Rcw, r.tcw, INTR)
    assert np.all(st > 0)
    for n, t in enumerate(r.tracks):
        assert np.abs(X[n] - t["X"]).max() < 1e-10
        assert np.nonzero(kept[off[n]:off[n + 1]])[0].tolist() == sorted(t["inliers"].tolist())


def _adapter_tracks(r, matches, depth, Rcw, tcw, directory):
    """include/lvba_adapter.hpp:build_tracks_and_fuse_with on the CPU (tests/adapter_tracks_emul.cpp).  Returns (off, obs, inlier
    flags, X, obs_to_track)."""
    import ctypes
    import subprocess
    from conftest import ROOT
    so = os.path.join(str(directory), "libadapter_emul.so")
    libdir = os.path.join(ROOT, "global-lvba_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", os.path.join(ROOT, "tests", "adapter_tracks_emul.cpp"),
                           "-o", so, "-L", libdir, "-llvba_hip", f"-Wl,-rpath,{libdir}"])
    ad = ctypes.CDLL(so)
    nk = np.array([len(k) for k in r.d["kps"]], np.int32)
    kp_xy = np.concatenate([np.asarray(k, np.float32)[:, :2] for k in r.d["kps"]]).astype(np.float32)
    pairs = np.array(r.d["pairs"], np.int32)
    moff = np.concatenate([[0], np.cumsum([len(m) for m in matches])]).astype(np.int64)
    mm = np.concatenate(matches).astype(np.int32)
    K = int(nk.sum())
    tl, X, obs, inl, o2t = np.zeros(K, np.int32), np.zeros((K, 3)), np.zeros((K, 2), np.int32), np.zeros(K, np.uint8), np.zeros(K, np.int32)
    depth = np.ascontiguousarray(depth, np.float32)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    ad.adapter_build_tracks.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 2 + \
        [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_double, ctypes.c_double] + [ctypes.c_void_p] * 5
    Rcw, tcw, intr = np.ascontiguousarray(Rcw), np.ascontiguousarray(tcw), np.ascontiguousarray(INTR)
    n = ad.adapter_build_tracks(len(nk), P(nk), P(kp_xy), len(pairs), P(pairs), P(moff), P(mm), P(depth), W, H, P(Rcw), P(tcw), P(intr),
                                3, 8.0, 3.0, P(tl), P(X), P(obs), P(inl), P(o2t))
    return np.concatenate([[0], np.cumsum(tl[:n])]), obs, inl, X, o2t, nk


def _same_tracks(ref_tracks, off, obs_img, obs_kp, kept, X):
    assert len(off) - 1 == len(ref_tracks)
    for n, t in enumerate(ref_tracks):
        a, b = int(off[n]), int(off[n + 1])
        assert np.array_equal(t["obs"][:, 0], obs_img[a:b]) and np.array_equal(t["obs"][:, 1], obs_kp[a:b])
        assert np.abs(X[n] - t["X"]).max() < 1e-10
        assert np.nonzero(kept[a:b])[0].tolist() == sorted(t["inliers"].tolist())


def test_product_host_loops_reproduce_the_reference_tracks(run, tmp_path_factory):
    """The product's two host-side track loops -- global-lvba_amd/pipeline.py:build_tracks_and_fuse (Python mirror) and
    include/lvba_adapter.hpp:build_tracks_and_fuse_with (the C++ binding) -- with the device code of fusion_device.h compiled for
    the host in the place of the GPU call: the reference's tracks, in its order, retries included."""
    import ctypes
    import subprocess
    import test_tracks_host as th
    from conftest import ROOT
    r = run
    lib = th.build_emul(tmp_path_factory.mktemp("emul_tracks_loop"))
    T = r.pipe.build_tracks_and_fuse(r.d["kps"], r.d["pairs"], r.d["matches"],
                                     lambda o, i, u: th._fuse(lib, o, i, u, r.depth, r.Rcw, r.tcw, INTR))
    _same_tracks(r.tracks, T["obs_off"], T["obs_img"], T["obs_kp"], T["kept"], T["X"])
    assert (T["attempts"] > 0).sum() >= 1 and (T["component_status"] > 0).sum() == len(r.tracks)
    # the C++ adapter
    off, obs, inl, X, o2t, nk = _adapter_tracks(r, r.d["matches"], r.depth, r.Rcw, r.tcw, tmp_path_factory.mktemp("adapter_emul"))
    K = int(nk.sum())
    _same_tracks(r.tracks, off, obs[:, 0], obs[:, 1], inl, X)
    # obs_to_track as the reference leaves it: the track of every member, -1 elsewhere
    want = -np.ones(K, np.int32)
    base = np.concatenate([[0], np.cumsum(nk)])
    for ti, t in enumerate(r.tracks):
        want[base[t["obs"][:, 0]] + t["obs"][:, 1]] = ti
    assert np.array_equal(o2t, want)


def test_problem_handed_to_ceres(run):
    r = run
    P = r.problem
    assert P is not None and P["n_cams"] == 12 and P["max_iter"] == 50 and P["linear_solver"] == 3          # DENSE_SCHUR
    assert P["q_const"].tolist() == [1] + [0] * 11 and P["t_const"].tolist() == [1] + [0] * 11              # camera 0 fixed
    assert np.all(P["q_tangent"] == 3) and np.all(P["loss_a"] == 0)                                         # manifold; loss = nullptr
    assert np.abs(P["q0"] - r.pipe.rot_to_quat_wxyz(r.Rcw)).max() < 1e-15 and np.array_equal(P["t0"], r.tcw)
    # landmarks: the usable tracks that find a plane in the map of the anchor clouds rebuilt from the refined poses
    L = r.loaded
    ap, ac = wo.merge_anchors(L["clouds"], r.x_opt, CFG["window_size"], CFG["anchor_leaf"])
    smap, _ = vo.build(ac, ap, CFG["stage_voxel_size"][1], np.float32(CFG["stage_eigen_ratio"][1]))
    mine = fo.build_tracks_and_fuse(r.d["kps"], r.d["pairs"], r.d["matches"], r.depth, r.Rcw, r.tcw, INTR)
    planes = [vo.find_plane(smap, t["X"], CFG["stage_voxel_size"][1]) for t in mine]
    keep = [i for i, p in enumerate(planes) if p is not None]
    assert len(keep) == P["n_points"] and 50 < len(keep) < len(mine)
    X = np.array([mine[i]["X"] for i in keep])
    assert np.abs(X - P["X0"]).max() < 1e-10
    pl = np.array([np.concatenate([planes[i][0], [planes[i][1]]]) for i in keep])
    sgn = np.sign(np.einsum("ij,ij->i", pl[:, :3], P["plane"][:, :3]))                                     # eigenvector sign is free
    assert np.abs(pl * sgn[:, None] - P["plane"]).max() < 1e-9
    # residual blocks: per landmark its distinct inlier observations, then its plane
    obs_off, obs_cam, obs_uv, k = [0], [], [], 0
    for li, i in enumerate(keep):
        t = mine[i]
        ref_inl = r.tracks[i]["inliers"]
        for idx in ref_inl:
            assert P["kind"][k] == 2 and P["point"][k] == li and P["cam"][k] == t["obs"][idx, 0]
            obs_cam.append(int(t["obs"][idx, 0])); obs_uv.append(r.d["kps"][t["obs"][idx, 0]][t["obs"][idx, 1]][:2].astype(np.float64))
            k += 1
        assert P["kind"][k] == 1 and P["point"][k] == li
        k += 1
        obs_off.append(len(obs_cam))
    assert k == len(P["kind"])
    prob = vis.VisualProblem(q=P["q0"], t=P["t0"], X=X, obs_off=np.array(obs_off), obs_cam=np.array(obs_cam, np.int32),
                             obs_uv=np.array(obs_uv).reshape(-1, 2), plane=pl, valid=np.ones(len(X), np.uint8), intr=INTR)
    O = vis.VisualOracle(prob)
    c0 = O.cost(*O.state())
    assert abs(c0 - P["cost0"]) < 1e-10 * P["cost0"]                               # sigma_px 0.5, sigma_plane 0.01, no robust loss
    # the write-back after the solve (:1651-1667): a solution installed in place of Ceres' is what the members hold afterwards
    rng = np.random.default_rng(3)
    q = P["q0"] + 1e-3 * rng.standard_normal(P["q0"].shape)
    t = P["t0"] + 1e-2 * rng.standard_normal(P["t0"].shape)
    Xs = P["X0"] + 1e-2 * rng.standard_normal(P["X0"].shape)
    assert r.S.optimize(solution=(q, t, Xs)) is not None
    Rn, tn = r.S.cam_poses(True)
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    assert np.abs(r.pipe.quat_wxyz_to_rot(qn) - Rn).max() < 1e-14 and np.array_equal(tn, t)


@pytest.mark.parametrize("variant", ["no_window", "odometry_anchors", "stage2_only", "ragged_windows", "image_stride"])
def test_run_lidar_ba_variants(run, variant):
    """The switches of runLidarBA / runWindowBA / DatasetIO (config keys window_ba/*, BALM_stage1/enable,
    data_config/image_sample_step) against the same switches of the restatements."""
    r = run
    L = r.loaded
    p = reference_params(r.tp)
    kw = dict(window_enable=True, window_size=CFG["window_size"], anchor_leaf=CFG["anchor_leaf"], use_rel=True, stage1_enable=True,
              stage_voxel_size=CFG["stage_voxel_size"], stage_eigen_ratio=CFG["stage_eigen_ratio"],
              # the plane thresholds recut() uses are a process-wide global of the reference (eigen_value_array, set per stage by
              # set_eigen_ratio_array, bavoxel.hpp): the window BA of a second LvbaSystem in the same process sees what the last
              # stage of the first one (the `run` fixture) left there, not the built-in defaults a fresh process starts with
              window_eigen_ratio=CFG["stage_eigen_ratio"][1])
    if variant == "no_window":                       # :221-229: every scan is its own anchor
        p["window_ba/enable"] = False; kw["window_enable"] = False
    elif variant == "odometry_anchors":              # :268-279: window BA only thins the clouds, poses stay the odometry's
        p["window_ba/use_window_ba_rel"] = False; kw["use_rel"] = False
    elif variant == "stage2_only":
        p["BALM_stage1/enable"] = False; kw["stage1_enable"] = False
    elif variant == "ragged_windows":                # 12 scans in windows of 5, 5, 2
        p["window_ba/size"] = 5; kw["window_size"] = 5
    elif variant == "image_stride":
        p["data_config/image_sample_step"] = 5
    S = rs.ReferenceSystem(r.root, p)
    try:
        if variant == "image_stride":                # handleImages / handleCamPoses: every 5th image and every 5th pose line
            ids = r.pipe.list_image_ids(os.path.join(r.root, "all_image"), 5)
            assert S.n_images == 3 and np.array_equal(S.image_ids(), ids)
            _, img = r.ds.load_poses_tum(os.path.join(r.root, "all_image", "image_poses.txt"), 5)
            Ri, ti = S.image_poses()
            assert np.abs(img[:, :9].reshape(-1, 3, 3) - Ri).max() < 1e-15 and np.array_equal(img[:, 9:], ti)
            return
        S.init()
        aidx, relR, relp = S.run_lidar_ba()
        R1, p1, _ = S.scan_poses()
    finally:
        S.close()
    out, rep = wo.run_lidar_ba(L["clouds"], L["poses"], **kw)
    assert np.abs(out[:, :9].reshape(-1, 3, 3) - R1).max() < 1e-9 and np.abs(out[:, 9:] - p1).max() < 1e-9
    assert np.abs(p1 - r.p0).max() > 0.01
    if variant == "no_window":
        assert aidx.tolist() == list(range(12)) and rep["n_anchors"] == 12
    if variant == "ragged_windows":
        assert aidx.tolist() == [0] * 5 + [1] * 5 + [2] * 2


def test_colmap_text_export(run, tmp_path):
    """The result writers: Colmap/sparse/images.txt and points3D.txt as LvbaSystem::VisualizeOptComparison
    writes them (src/lvba_system.cpp:2018-2024, :2126-2137) against dataset.write_images_txt / write_points3d_txt.  The
    reference colours its LiDAR map from the images; with no image codec the stand-in's imread hands out the pattern
    (b, g, r) = (x, y, x + y) mod 256, so a point's colour tells the pixel it was drawn from."""
    r = run
    S = rs.ReferenceSystem(r.root, reference_params(r.tp))
    try:
        S.init()
        S.build_grid_map(); S.update_camera_poses(); S.generate_depth(W, H)     # fills Rcw_all_optimized_ (no LiDAR BA: = odometry)
        Rcw, tcw = S.cam_poses(True)
        S.export_colmap(W, H)
    finally:
        S.close()
    sparse = os.path.join(r.root, "Colmap", "sparse")
    mine = tmp_path / "images.txt"
    r.ds.write_images_txt(str(mine), r.pipe.rot_to_quat_wxyz(Rcw), tcw)
    assert open(os.path.join(sparse, "images.txt")).read() == mine.read_text()
    lines = open(os.path.join(sparse, "points3D.txt")).read().splitlines()
    assert len(lines) > 1000
    rows = np.array([[float(v) for v in ln.split()] for ln in lines])
    assert np.array_equal(rows[:, 0], np.arange(len(rows))) and np.all(rows[:, 7] == 0)
    mine_p = tmp_path / "points3D.txt"
    r.ds.write_points3d_txt(str(mine_p), rows[:, 1:4], rows[:, 4:7])
    assert mine_p.read_text().splitlines() == lines                                # same number formatting
    # colours: r = (b + g) mod 256 for every point, and b, g are the pixel some camera sees the point at
    rr, gg, bb = rows[:, 4].astype(int), rows[:, 5].astype(int), rows[:, 6].astype(int)
    assert np.array_equal(rr, (bb + gg) % 256)
    from oracle import track_oracle as to
    hits = 0
    for p, g, b in zip(rows[:200, 1:4], gg[:200], bb[:200]):
        for m in range(len(Rcw)):
            uv = to.project(INTR, Rcw[m], tcw[m], p)
            if uv is not None and int(round(uv[0])) % 256 == b and int(round(uv[1])) % 256 == g:
                hits += 1
                break
    assert hits >= 190                                                             # 6-decimal text coordinates: a few sit on a pixel edge


@pytest.mark.parametrize("obser_thr,angle,thr", [(4, 5.0, 2.0), (3, 15.0, 1.0), (5, 2.0, 6.0)])
def test_track_fusion_thresholds(run, tmp_path_factory, obser_thr, angle, thr):
    """obser_thr_ / track_fusion/min_view_angle / track_fusion/reproj_mean_thr away from their defaults, on the unrefined
    (odometry) cameras: the reference's tracks against the oracle and against the product's loop + device code on the host."""
    import test_tracks_host as th
    r = run
    S = rs.ReferenceSystem(r.root, reference_params(r.tp))
    try:
        S.init()
        S.build_grid_map(); S.update_camera_poses()
        depth = S.generate_depth(W, H)
        Rcw, tcw = S.cam_poses(True)
        S.set_features(r.d["kps"], {pr: m for pr, m in zip(r.d["pairs"], r.d["matches"])})
        tracks = S.build_tracks(obser_thr, angle, thr)
    finally:
        S.close()
    mine = fo.build_tracks_and_fuse(r.d["kps"], r.d["pairs"], r.d["matches"], depth, Rcw, tcw, INTR, obser_thr=obser_thr,
                                    min_view_angle_deg=angle, reproj_thr=thr)
    assert len(mine) == len(tracks) > 20
    for a, b in zip(tracks, mine):
        assert np.array_equal(a["obs"], b["obs"]) and np.abs(a["X"] - b["X"]).max() < 1e-10
        assert sorted(a["inliers"].tolist()) == np.nonzero(b["kept"])[0].tolist()
    lib = th.build_emul(tmp_path_factory.mktemp("emul_thr"))
    T = r.pipe.build_tracks_and_fuse(r.d["kps"], r.d["pairs"], r.d["matches"],
                                     lambda o, i, u: th._fuse(lib, o, i, u, depth, Rcw, tcw, INTR, obser_thr, angle, thr), obser_thr)
    _same_tracks(tracks, T["obs_off"], T["obs_img"], T["obs_kp"], T["kept"], T["X"])


def test_tracks_with_false_matches(run, tmp_path_factory):
    """Wrong matches chain tracks together: components with several key points of one image, depth points far apart, many
    fusion failures and therefore many retries from later members -- reference against oracle and product loop."""
    import test_tracks_host as th
    r = run
    rng = np.random.default_rng(17)
    matches = []
    for (i, j), m in zip(r.d["pairs"], r.d["matches"]):
        extra = np.stack([rng.integers(0, len(r.d["kps"][i]), 6), rng.integers(0, len(r.d["kps"][j]), 6)], 1)
        matches.append(np.concatenate([np.asarray(m, np.int64), extra]))
    S = rs.ReferenceSystem(r.root, reference_params(r.tp))
    try:
        S.init()
        S.build_grid_map(); S.update_camera_poses()
        depth = S.generate_depth(W, H)
        Rcw, tcw = S.cam_poses(True)
        S.set_features(r.d["kps"], {pr: m for pr, m in zip(r.d["pairs"], matches)})
        tracks = S.build_tracks()
    finally:
        S.close()
    assert any(len(set(t["obs"][:, 0].tolist())) < len(t["obs"]) for t in tracks)      # some track sees an image twice
    mine = fo.build_tracks_and_fuse(r.d["kps"], r.d["pairs"], matches, depth, Rcw, tcw, INTR)
    assert len(mine) == len(tracks) > 20
    for a, b in zip(tracks, mine):
        assert np.array_equal(a["obs"], b["obs"]) and np.abs(a["X"] - b["X"]).max() < 1e-10
        assert sorted(a["inliers"].tolist()) == np.nonzero(b["kept"])[0].tolist()
    lib = th.build_emul(tmp_path_factory.mktemp("emul_noise"))
    T = r.pipe.build_tracks_and_fuse(r.d["kps"], r.d["pairs"], matches, lambda o, i, u: th._fuse(lib, o, i, u, depth, Rcw, tcw, INTR))
    _same_tracks(tracks, T["obs_off"], T["obs_img"], T["obs_kp"], T["kept"], T["X"])
    assert (T["attempts"] > 0).sum() >= 3 and T["attempts"].max() >= 2
    off, obs, inl, X, _, _ = _adapter_tracks(r, matches, depth, Rcw, tcw, tmp_path_factory.mktemp("adapter_noise"))
    _same_tracks(tracks, off, obs[:, 0], obs[:, 1], inl, X)


def test_camera_update_edge_cases(tmp_path):
    """updateCameraPosesFromLidar (:412-446) where the nearest-scan search leaves the scan range: images before the first scan,
    after the last one, exactly between two scans (lower_bound, the earlier scan only if STRICTLY closer), and more images than
    a 1:1 pairing would need."""
    ds = importlib.import_module("global-lvba_amd.dataset")
    pipe = importlib.import_module("global-lvba_amd.pipeline")
    import test_gpu_pipeline as tp
    rng = np.random.default_rng(4)
    n = 5
    times = 100.0 + 0.5 * np.arange(n)
    img_t = np.array([99.2, 100.0, 100.25, 100.26, 100.74, 101.5, 102.0, 103.7])
    def rand_pose():
        q = rng.standard_normal(4); q /= np.linalg.norm(q)
        return np.concatenate([np.asarray(ds.quat_to_rot(*q)).reshape(-1), rng.standard_normal(3)])
    d = dict(times=times, img_t=img_t, clouds=[rng.standard_normal((50, 3)).astype(np.float32) for _ in range(n)],
             odo=np.array([rand_pose() for _ in range(n)]))
    root = str(tmp_path / "seq")
    os.makedirs(os.path.join(root, "all_pcd_body")); os.makedirs(os.path.join(root, "all_image"))
    for t, c in zip(times, d["clouds"]):
        ds.save_pcd(os.path.join(root, "all_pcd_body", f"{t:.6f}.pcd"), np.concatenate([c, np.zeros((len(c), 1), np.float32)], 1), mode="ascii")
    ds.write_poses_tum(os.path.join(root, "all_pcd_body", "lidar_poses.txt"), times, d["odo"])
    for t in img_t:
        open(os.path.join(root, "all_image", f"{t:.6f}.jpg"), "wb").close()
    cam = np.array([rand_pose() for _ in img_t])
    ds.write_poses_tum(os.path.join(root, "all_image", "image_poses.txt"), img_t, cam)
    S = rs.ReferenceSystem(root, reference_params(tp))
    try:
        assert S.n_scans == n and S.n_images == len(img_t)
        for i in range(n):                                                         # the ascii PCD branch of the stand-in reader
            assert np.array_equal(S.cloud(i)[:, :3], d["clouds"][i])
        S.init()
        R0, p0, ts = S.scan_poses()
        x_orig = np.concatenate([R0.reshape(-1, 9), p0], 1)
        x_opt = np.array([rand_pose() for _ in range(n)])                          # "refined" poses: anything
        S.set_scan_poses(x_opt[:, :9].reshape(-1, 3, 3), x_opt[:, 9:])
        Rn, tn = S.update_camera_poses()
        _, cam_in = ds.load_poses_tum(os.path.join(root, "all_image", "image_poses.txt"), 1)
        ids = S.image_ids()
    finally:
        S.close()
    assert np.array_equal(ids, img_t)
    got = pipe.update_camera_poses_from_lidar(x_opt, x_orig, ts, ids, cam_in)
    assert np.abs(got[:, :9].reshape(-1, 3, 3) - Rn).max() < 1e-13 and np.abs(got[:, 9:] - tn).max() < 1e-12


def test_colmap_database_import(tmp_path):
    """LvbaSystem::loadFromColmapDB (:510-685, the system's libsqlite3 linked) against dataset.load_colmap_db on a database with
    the cases the reader has branches for: image ids that do not follow the time order (pairs stored the other way round),
    4- and 6-column key points, a missing key point blob, a blob of the wrong size, out-of-range match indices, a pair with
    cols != 2, a pair that is absent."""
    import sqlite3
    import test_gpu_pipeline as tp
    ds = importlib.import_module("global-lvba_amd.dataset")
    rng = np.random.default_rng(9)
    n = 5
    times = 10.0 + 0.5 * np.arange(n)
    img_t = times + 0.01
    root = str(tmp_path / "seq")
    os.makedirs(os.path.join(root, "all_pcd_body")); os.makedirs(os.path.join(root, "all_image"))
    I12 = np.concatenate([np.eye(3).reshape(-1), np.zeros(3)])
    for t in times:
        ds.save_pcd(os.path.join(root, "all_pcd_body", f"{t:.6f}.pcd"), rng.standard_normal((20, 4)).astype(np.float32))
    ds.write_poses_tum(os.path.join(root, "all_pcd_body", "lidar_poses.txt"), times, np.tile(I12, (n, 1)))
    for t in img_t:
        open(os.path.join(root, "all_image", f"{t:.6f}.png"), "wb").close()
    ds.write_poses_tum(os.path.join(root, "all_image", "image_poses.txt"), img_t, np.tile(I12, (n, 1)))
    db_id = [4, 2, 9, 1, 7]                                                     # database image ids, not in time order
    nk = [30, 25, 0, 40, 35]
    con = sqlite3.connect(os.path.join(root, "colmap.db"))
    con.execute("CREATE TABLE images (image_id INTEGER PRIMARY KEY, name TEXT)")
    con.execute("CREATE TABLE keypoints (image_id INTEGER PRIMARY KEY, rows INTEGER, cols INTEGER, data BLOB)")
    con.execute("CREATE TABLE two_view_geometries (pair_id INTEGER PRIMARY KEY, rows INTEGER, cols INTEGER, data BLOB)")
    for i, t in enumerate(img_t):
        con.execute("INSERT INTO images VALUES (?, ?)", (db_id[i], f"{t:.6f}.png"))
        cols = 6 if i == 1 else 4
        kp = rng.uniform(0, 400, (nk[i], cols)).astype(np.float32)
        if i == 2:
            continue                                                            # image 2: no key point row at all
        blob = kp.tobytes()[:-4] if i == 4 else kp.tobytes()                    # image 4: blob one float short -> ignored
        con.execute("INSERT INTO keypoints VALUES (?, ?, ?, ?)", (db_id[i], nk[i], cols, blob))
    def put(i, j, m, cols=2):
        a, b = db_id[i], db_id[j]
        m = np.asarray(m, np.uint32)
        if a > b:                                                               # COLMAP stores (smaller id, larger id)
            m = m[:, ::-1]
        con.execute("INSERT INTO two_view_geometries VALUES (?, ?, ?, ?)", (ds.image_ids_to_pair_id(a, b), len(m), cols, np.ascontiguousarray(m).tobytes()))
    put(0, 1, [[0, 1], [5, 6], [29, 24], [30, 3], [2, 25]])                     # two out-of-range rows
    put(0, 3, [[1, 2], [3, 39], [7, 7]])                                        # ids 4 > 1: stored swapped
    put(1, 3, [[4, 4], [24, 0]])
    put(0, 2, [[0, 0]])                                                         # image 2 has no key points
    put(3, 4, [[0, 0]])                                                         # image 4's blob was rejected
    con.execute("INSERT INTO two_view_geometries VALUES (?, ?, ?, ?)", (ds.image_ids_to_pair_id(db_id[1], db_id[4]) + 10**6, 1, 3, b"\\0" * 12))
    con.commit(); con.close()
    p = reference_params(tp); p["data_config/colmap_db_path"] = "colmap.db"
    S = rs.ReferenceSystem(root, p)
    try:
        S.init()
        ok, kps, matches = S.load_colmap_db()
    finally:
        S.close()
    assert ok
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    names = [f"{t:.6f}.png" for t in img_t]
    mk, mm = ds.load_colmap_db(os.path.join(root, "colmap.db"), names, pairs)
    for i in range(n):
        assert len(kps[i]) == len(mk[i]) == (0 if i in (2, 4) else nk[i])
        if len(mk[i]):
            assert np.array_equal(kps[i], mk[i][:, :4])                         # x y sigma extremum (cols 5, 6 of a 6-column blob unused)
    for pr, m in zip(pairs, mm):
        assert np.array_equal(matches[pr], m), pr
    assert len(matches[(0, 1)]) == 3 and matches[(0, 3)].tolist() == [[1, 2], [3, 39], [7, 7]] and len(matches[(0, 2)]) == 0
    # a database whose image count differs is refused (:545-552)
    con = sqlite3.connect(os.path.join(root, "colmap.db")); con.execute("DELETE FROM images WHERE image_id=9"); con.commit(); con.close()
    S = rs.ReferenceSystem(root, p)
    try:
        S.init()
        assert S.load_colmap_db()[0] is False
    finally:
        S.close()


def test_adapter_track_loop_without_matches(run, tmp_path_factory):
    """lvba::build_tracks_and_fuse_with on an empty match table: no batch, no tracks, obs_to_track all -1."""
    r = run
    empty = [np.zeros((0, 2), np.int64) for _ in r.d["pairs"]]
    off, obs, inl, X, o2t, nk = _adapter_tracks(r, empty, r.depth, r.Rcw, r.tcw, tmp_path_factory.mktemp("adapter_empty"))
    assert off.tolist() == [0] and np.all(o2t == -1)
