"""Host-side mirror of the reference's top-level flow over the C-ABI: everything between "a dataset in memory" and "refined
LiDAR poses, camera poses and landmarks", with every compute step in liblvba_hip.so on the GPU.

    LvbaSystem::runFullPipeline               src/lvba_system.cpp:136-142   initFromDatasetIO -> runLidarBA -> visual stage
    LvbaSystem::runVisualBAWithLidarAssist    src/lvba_system.cpp:144-154   grid map, camera poses from the refined LiDAR
                                                                            poses, depth images, (features), tracks + fusion,
                                                                            optimizeCameraPoses
    LvbaSystem::updateCameraPosesFromLidar    src/lvba_system.cpp:412-446   T_cam_new = (T_opt T_orig^-1) cam_orig of the
                                                                            nearest scan in time
    camera extrinsics                         src/lvba_system.cpp:860-869   Rcw = Rci Rwi^T, tcw = -Rcw Pwi + tci
    anchor clouds + plane map of the visual stage  src/lvba_system.cpp:1453-1507

Out of scope, as in DESIGN.md: SIFT extraction / matching (SiftGPU, `extractAndMatchFeaturesGPU`) -- keypoints and inlier
matches are inputs here, e.g. from a COLMAP database through dataset.load_colmap_db, which is the reference's own alternative
(`loadFromColmapDB`); ROS publishing and the OpenCV visualisations.
"""
from __future__ import annotations

import numpy as np

from . import visual as V
from .voxel import Scans

DEFAULTS = dict(                                   # config/config.yaml of the reference
    window_enable=True, window_size=20, anchor_leaf=0.01, use_rel=True,
    stage1_enable=True, stage_voxel_size=(1.0, 0.5), stage_eigen_ratio=((0.2,) * 4, (0.08,) * 4),
    obser_thr=3, min_view_angle_deg=8.0, reproj_mean_thr_px=3.0, depth_half_window_s=0.5, depth_voxel=0.5,
    sigma_px=0.5, sigma_plane=0.01)


def _mat(T):
    T = np.asarray(T, np.float64).reshape(-1, 12)
    return T[:, :9].reshape(-1, 3, 3), T[:, 9:]


def update_camera_poses_from_lidar(x_opt, x_orig, scan_times, image_times, cam_orig):
    """src/lvba_system.cpp:412-446.  Poses are [n,12] (R row-major, p), T_world<-imu.  For every image the scan nearest in
    time (std::lower_bound, the previous one if it is strictly closer) gives T_delta = T_opt T_orig^-1; the image pose
    becomes T_delta o cam_orig."""
    Ro, po = _mat(x_opt)
    Rb, pb = _mat(x_orig)
    Rc, pc = _mat(cam_orig)
    ts = np.asarray(scan_times, np.float64)
    out = np.zeros((len(Rc), 12))
    for i, t in enumerate(np.asarray(image_times, np.float64)):
        it = int(np.searchsorted(ts, t, side="left"))
        idx = len(ts) - 1 if it == len(ts) else it
        if 0 < it < len(ts) and abs(ts[idx - 1] - t) < abs(ts[idx] - t):
            idx -= 1
        if idx >= len(Ro) or idx >= len(Rb):
            out[i, :9], out[i, 9:] = Rc[i].reshape(-1), pc[i]
            continue
        Rd = Ro[idx] @ Rb[idx].T                                            # T_opt * T_orig.inverse()
        pd = po[idx] - Rd @ pb[idx]
        out[i, :9] = (Rd @ Rc[i]).reshape(-1)
        out[i, 9:] = Rd @ pc[i] + pd
    return out


def camera_from_imu(T_wi, Rci, tci):
    """Rcw = Rci Rwi^T, tcw = -Rcw Pwi + tci (src/lvba_system.cpp:860-861)."""
    R, p = _mat(T_wi)
    Rci, tci = np.asarray(Rci, np.float64).reshape(3, 3), np.asarray(tci, np.float64).reshape(3)
    Rcw = np.einsum("ij,nkj->nik", Rci, R)
    tcw = -np.einsum("nij,nj->ni", Rcw, p) + tci
    return Rcw, tcw


def rot_to_quat_wxyz(R):
    """Eigen::Quaterniond(R).normalize() as [w, x, y, z] (src/lvba_system.cpp:1514-1517)."""
    from .dataset import rot_to_quat
    return np.array([rot_to_quat(r) for r in np.asarray(R).reshape(-1, 3, 3)])


def quat_wxyz_to_rot(q):
    from .dataset import quat_to_rot
    return np.array([quat_to_rot(*qq) for qq in np.asarray(q).reshape(-1, 4)])


def match_graph(n_keypoints, pairs, matches):
    """Adjacency of the key point match graph as BuildTracksAndFuse3D builds it (src/lvba_system.cpp:932-952): pairs visited in
    pairIndex order (i < j, i-major), every match appended to both ends' lists.  adj[i] = {key point: [(image, key point), ..]}."""
    N = len(n_keypoints)
    adj = [dict() for _ in range(N)]
    todo = []
    for (i, j), m in zip(pairs, matches):
        m = np.asarray(m, np.int64).reshape(-1, 2)
        if i > j:
            i, j, m = j, i, m[:, ::-1]
        todo.append((i, j, m))
    todo.sort(key=lambda q: (q[0], q[1]))
    for i, j, m in todo:
        for ki, kj in m:
            if ki < 0 or kj < 0 or ki >= n_keypoints[i] or kj >= n_keypoints[j]:
                continue
            adj[i].setdefault(int(ki), []).append((j, int(kj)))
            adj[j].setdefault(int(kj), []).append((i, int(ki)))
    return adj


def bfs_order(adj, start):
    """The BFS of :962-981 from `start` = (image, key point) over a whole connected component."""
    from collections import deque
    seen, comp, q = {start}, [], deque([start])
    while q:
        ci, ck = q.popleft()
        comp.append((ci, ck))
        for nb in adj[ci].get(ck, ()):
            if nb not in seen:
                seen.add(nb)
                q.append(nb)
    return comp


def match_components(n_keypoints, pairs, matches, obser_thr=3):
    """Connected components of the match graph that pass the two size checks (:983-1014: >= obser_thr observations from >=
    obser_thr images).  Returns (adj, comps); comps[c] = members sorted in scan order (image, key point): the reference starts
    its BFS at comps[c][0] and, if the fusion drops the component, again at comps[c][1], comps[c][2], ..."""
    adj = match_graph(n_keypoints, pairs, matches)
    seen = [set() for _ in adj]
    comps = []
    for i in range(len(adj)):
        for ki in sorted(adj[i]):
            if ki in seen[i]:
                continue
            comp = bfs_order(adj, (i, ki))
            for ci, ck in comp:
                seen[ci].add(ck)
            if len(comp) >= obser_thr and len({c for c, _ in comp}) >= obser_thr:
                comps.append(sorted(comp))
    return adj, comps


def build_components(n_keypoints, pairs, matches, obser_thr=3):
    """First-attempt BFS order of every component of match_components as CSR arrays (obs_off, obs_img, obs_kp)."""
    adj, comps = match_components(n_keypoints, pairs, matches, obser_thr)
    off, img, kp = [0], [], []
    for members in comps:
        for ci, ck in bfs_order(adj, members[0]):
            img.append(ci); kp.append(ck)
        off.append(len(img))
    return np.asarray(off, np.int64), np.asarray(img, np.int32), np.asarray(kp, np.int32)


def build_tracks_and_fuse(keypoints, pairs, matches, fuse_fn, obser_thr=3):
    """The track loop of BuildTracksAndFuse3D (src/lvba_system.cpp:954-1246) with the per-component fusion batched:
    fuse_fn(obs_off, obs_img, obs_uv) -> (status, X, err, kept) is lvba_fuse_tracks on the GPU.  A component the fusion drops is
    released by the reference (:1197, :1203) and met again at its next member in scan order, i.e. fused again in another BFS
    order; round r of the loop below fuses, in one batch, the r-th attempt of every component that is still dropped.  Tracks
    come out in the reference's order (by the key point their successful BFS started from).
    Returns dict(obs_off, obs_img, obs_kp, obs_uv, kept: CSR arrays of the tracks; X, err, status: per track; component_status:
    per component (0 = dropped after all attempts), attempts: per track, 0-based)."""
    nk = [len(k) for k in keypoints]
    adj, comps = match_components(nk, pairs, matches, obser_thr)
    comp_status = np.zeros(len(comps), np.uint8)
    done = []                                                               # (start, order, X, err, status, kept, attempt)
    pending, attempt = list(range(len(comps))), 0
    while pending:
        orders = [bfs_order(adj, comps[c][attempt]) for c in pending]
        off = np.concatenate([[0], np.cumsum([len(o) for o in orders])]).astype(np.int64)
        flat = [ob for o in orders for ob in o]
        img = np.array([i for i, _ in flat], np.int32)
        uv = np.array([keypoints[i][k][:2] for i, k in flat], np.float32).reshape(-1, 2)
        status, X, err, kept = fuse_fn(off, img, uv)
        nxt = []
        for n, c in enumerate(pending):
            if status[n]:
                comp_status[c] = status[n]
                done.append((comps[c][attempt], orders[n], X[n].copy(), float(err[n]), int(status[n]),
                             np.asarray(kept[off[n]:off[n + 1]]).copy(), attempt))
            elif attempt + 1 < len(comps[c]):
                nxt.append(c)
        pending, attempt = nxt, attempt + 1
    done.sort(key=lambda d: d[0])
    off = np.concatenate([[0], np.cumsum([len(d[1]) for d in done])]).astype(np.int64)
    flat = [ob for d in done for ob in d[1]]
    img = np.array([i for i, _ in flat], np.int32)
    kp = np.array([k for _, k in flat], np.int32)
    uv = np.array([keypoints[i][k][:2] for i, k in flat], np.float32).reshape(-1, 2)
    return dict(obs_off=off, obs_img=img, obs_kp=kp, obs_uv=uv,
                kept=np.concatenate([d[5] for d in done]).astype(np.uint8) if done else np.zeros(0, np.uint8),
                X=np.array([d[2] for d in done]).reshape(-1, 3), err=np.array([d[3] for d in done]),
                status=np.array([d[4] for d in done], np.uint8), attempts=np.array([d[6] for d in done], np.int32),
                component_status=comp_status)


def run_visual_ba_with_lidar_assist(scans, x_opt, x_orig, scan_times, image_times, image_poses, Rci, tci, intr, width, height,
                                    keypoints, pairs, matches, **cfg):
    """LvbaSystem::runVisualBAWithLidarAssist (src/lvba_system.cpp:144-154) from the refined LiDAR poses to the refined
    cameras.  scans: a voxel.Scans holding the raw clouds; keypoints[i] = [n_i, 2] float pixel coordinates; pairs / matches as
    build_tracks takes them.  Returns a dict (cameras before / after, tracks, landmarks, planes, traces)."""
    c = dict(DEFAULTS); c.update(cfg)
    cam_new = update_camera_poses_from_lidar(x_opt, x_orig, scan_times, image_times, image_poses)      # poses_
    Rcw, tcw = camera_from_imu(cam_new, Rci, tci)                                                      # Rcw_all_optimized_
    Rcw0, tcw0 = camera_from_imu(image_poses, Rci, tci)                                                # Rcw_all_ (before)
    # generateDepthWithVoxel (+ buildGridMapFromOptimized)
    depth = V.DepthImages.render(scans, x_opt, scan_times, image_times, Rcw, tcw, intr, width, height,
                                 half_window_s=c["depth_half_window_s"], voxel_size=c["depth_voxel"])
    try:
        # BuildTracksAndFuse3D
        T = build_tracks_and_fuse(keypoints, pairs, matches,
                                  lambda o, i, u: V.fuse_tracks(o, i, u, Rcw, tcw, intr, depth=depth, obser_thr=c["obser_thr"],
                                                                min_view_angle_deg=c["min_view_angle_deg"],
                                                                reproj_mean_thr_px=c["reproj_mean_thr_px"]), c["obser_thr"])
    finally:
        depth.close()
    off, img, uv, kept, X, err = T["obs_off"], T["obs_img"], T["obs_uv"], T["kept"], T["X"], T["err"]
    status = T["component_status"]
    tr = np.arange(len(X))                                                   # tracks_: every one is usable (:1436-1441)
    out = dict(cam_poses=cam_new, Rcw_before=Rcw0, tcw_before=tcw0, Rcw_lidar=Rcw, tcw_lidar=tcw, track_status=status,
               n_components=len(status), tracks=T)
    if len(tr) == 0:
        out.update(Rcw=Rcw, tcw=tcw, landmarks=np.zeros((0, 3)), landmark_valid=np.zeros(0, np.uint8), trace=[], termination="NO_TRACKS")
        return out
    # anchor clouds from the refined poses and the plane map of the visual stage (:1453-1507)
    m = scans.window_ba(x_opt, window_size=c["window_size"], anchor_leaf=c["anchor_leaf"], merge_only=True)
    try:
        with m["anchor_scans"].voxel_map(m["anchor_poses"], c["stage_voxel_size"][1], np.float32(c["stage_eigen_ratio"][1])) as vmap:
            plane, pvalid = vmap.find_planes(X[tr])
    finally:
        m["anchor_scans"].close()
    # inlier observations of the usable tracks, one residual per distinct observation (:1612-1634)
    o_off, o_cam, o_uv = [0], [], []
    for t in tr:
        a, b = int(off[t]), int(off[t + 1])
        sel = np.nonzero(kept[a:b])[0] + a
        o_cam.extend(img[sel].tolist()); o_uv.extend(uv[sel].astype(np.float64).tolist())
        o_off.append(len(o_cam))
    q0 = rot_to_quat_wxyz(Rcw)
    (q, t, Xn), trace, term, rc, valid = V.optimize_camera_poses(
        q0, tcw, X[tr], np.asarray(o_off, np.int64), np.asarray(o_cam, np.int32), np.asarray(o_uv, np.float64).reshape(-1, 2),
        plane[:, :3], plane[:, 3], intr, c["sigma_px"], c["sigma_plane"])
    out.update(Rcw=quat_wxyz_to_rot(q), tcw=np.asarray(t), q=q, landmarks=np.asarray(Xn), landmarks_before=X[tr],
               landmark_valid=valid, track_ids=tr, plane=plane, plane_valid=pvalid, obs_off=np.asarray(o_off), obs_cam=np.asarray(o_cam),
               obs_uv=np.asarray(o_uv).reshape(-1, 2), trace=trace, termination=term, status=rc, mean_reproj=err[tr])
    return out


def run_full_pipeline(clouds, poses, scan_times, image_times, image_poses, Rci, tci, intr, width, height, keypoints, pairs,
                      matches, enable_lidar_ba=True, enable_visual_ba=True, device=0, **cfg):
    """LvbaSystem::runFullPipeline (src/lvba_system.cpp:136-142) on in-memory data: clouds = body-frame [n_i, >=3] float32
    arrays, poses [n,12] = x_buf_ (T_world<-imu), image_poses [m,12] the image poses from the odometry."""
    c = dict(DEFAULTS); c.update(cfg)
    x_orig = np.asarray(poses, np.float64).reshape(-1, 12).copy()
    out = dict(poses_before=x_orig)
    with Scans(clouds, device=device) as scans:
        x_opt = x_orig
        if enable_lidar_ba:
            x_opt, report = scans.lidar_ba(x_orig, window_enable=c["window_enable"], window_size=c["window_size"],
                                           anchor_leaf=c["anchor_leaf"], use_rel=c["use_rel"], stage1_enable=c["stage1_enable"],
                                           stage_voxel_size=c["stage_voxel_size"], stage_eigen_ratio=c["stage_eigen_ratio"])
            out["lidar_report"] = report
        out["poses"] = np.asarray(x_opt).reshape(-1, 12)
        if enable_visual_ba:
            out["visual"] = run_visual_ba_with_lidar_assist(scans, out["poses"], x_orig, scan_times, image_times, image_poses, Rci,
                                                            tci, intr, width, height, keypoints, pairs, matches, **c)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# The same flow from a dataset directory in the reference's on-disk layout (README.md "Dataset", src/dataset_io.cpp):
#     <data_path>/all_pcd_body/lidar_poses.txt + <timestamp>.pcd      TUM poses (T_world<-imu) + body-frame scans
#     <data_path>/all_image/image_poses.txt + <timestamp>.png         TUM poses of the images + the images (only their names are read)
#     <data_path>/<colmap_db_path>                                    keypoints + inlier matches (loadFromColmapDB)
# ---------------------------------------------------------------------------------------------------------------------
def list_image_ids(image_dir, stride=1):
    """DatasetIO::handleImages (src/dataset_io.cpp:77-131): numeric ids of *.png/.jpg/.jpeg/.bmp, sorted, every stride-th."""
    import os
    from .dataset import parse_timestamp_from_name
    ids = []
    for name in os.listdir(image_dir):
        if os.path.splitext(name)[1] not in (".png", ".jpg", ".jpeg", ".bmp"):
            continue
        t = parse_timestamp_from_name(name)
        if t is not None:
            ids.append(t)
    ids.sort()
    return np.asarray(ids[::max(1, int(stride))], np.float64)


def extrinsics_from_config(Rcl, Pcl, extrinsic_R, extrinsic_T):
    """Rci = Rcl Rli, tci = Rcl tli + tcl with (Rli, tli) the inverse of the lidar->imu extrinsic (src/lvba_system.cpp:484-504)."""
    Rcl, tcl = np.asarray(Rcl, np.float64).reshape(3, 3), np.asarray(Pcl, np.float64).reshape(3)
    Ril, til = np.asarray(extrinsic_R, np.float64).reshape(3, 3), np.asarray(extrinsic_T, np.float64).reshape(3)
    Rli = Ril.T
    tli = -Rli @ til
    return Rcl @ Rli, Rcl @ tli + tcl


def run_dataset(data_path, colmap_db_path, intr, width, height, Rcl, Pcl, extrinsic_R=np.eye(3), extrinsic_T=np.zeros(3),
                image_sample_step=1, out_dir=None, device=0, **cfg):
    """initFromDatasetIO + runFullPipeline on a dataset directory; with out_dir, the refined LiDAR poses (TUM) and the COLMAP
    text files images.txt / points3D.txt the reference writes (src/lvba_system.cpp:2018-2137) are saved there.  images.txt is
    the reference's, character for character (tests/test_ref_system.py); points3D.txt has the reference's format but holds
    the refined visual landmarks in white -- the reference fills it with its LiDAR map coloured from the images
    (VisualizeOptComparison), which needs an image codec and is visualisation, outside the scope contract."""
    import os
    from . import dataset as D
    ds = D.load_dataset(data_path)
    img_dir = os.path.join(data_path, "all_image")
    image_ids = list_image_ids(img_dir, image_sample_step)
    _, image_poses = D.load_poses_tum(os.path.join(img_dir, "image_poses.txt"), image_sample_step)
    if len(image_poses) != len(image_ids):
        raise ValueError(f"{len(image_ids)} images but {len(image_poses)} image poses")          # :457-460
    names = [f"{t:.6f}.png" for t in image_ids]                                                   # getImagePath: std::to_string
    pairs = [(i, j) for i in range(len(image_ids)) for j in range(i + 1, len(image_ids))]        # image_pairs_, :462-466
    kps, matches = D.load_colmap_db(colmap_db_path if os.path.isabs(colmap_db_path) else os.path.join(data_path, colmap_db_path),
                                    names, pairs)
    keep = [k for k, m in enumerate(matches) if len(m)]
    Rci, tci = extrinsics_from_config(Rcl, Pcl, extrinsic_R, extrinsic_T)
    out = run_full_pipeline([c[:, :3] for c in ds["clouds"]], ds["poses"], ds["timestamps"], image_ids, image_poses, Rci, tci, intr,
                            width, height, [k[:, :2] for k in kps], [pairs[k] for k in keep], [matches[k] for k in keep],
                            device=device, **cfg)
    out.update(image_ids=image_ids, scan_times=ds["timestamps"])
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        D.write_poses_tum(os.path.join(out_dir, "lidar_poses_refined.txt"), ds["timestamps"], out["poses"])
        v = out.get("visual")
        if v is not None and len(v.get("landmarks", [])):
            D.write_images_txt(os.path.join(out_dir, "images.txt"), rot_to_quat_wxyz(v["Rcw"]), v["tcw"])
            ok = v["landmark_valid"] > 0
            D.write_points3d_txt(os.path.join(out_dir, "points3D.txt"), v["landmarks"][ok], np.full((int(ok.sum()), 3), 255))
    return out
