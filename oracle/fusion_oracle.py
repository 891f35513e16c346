"""CPU oracle of the LiDAR-assisted landmark initialisation.  TEST INFRASTRUCTURE ONLY.

Restates (paths relative to /root/reference)
  LvbaSystem::buildGridMapFromOptimized     src/lvba_system.cpp:1266-1338  world points of every scan hashed into 0.5 m voxels
                                                                           (float key quotient, -1 if negative); per image the
                                                                           voxels touched by the scans within +-0.5 s
  LvbaSystem::generateDepthWithVoxel        src/lvba_system.cpp:835-919    z-buffer of ALL grid-map points of those voxels:
                                                                           pixel = (int)u, (int)v, Z < 1e-3 skipped, float min
  LvbaSystem::BuildTracksAndFuse3D          src/lvba_system.cpp:1016-1225  per track: depth-fused candidate, triangulation
                                                                           candidate, selection by mean reprojection error
  fetchDepthBilinear / backProjectPixelDepthDistorted / camToWorld          include/utils.hpp:235-284
The helper functions of include/utils.hpp are pinned against the reference's own code (tests/test_ref_pin.py); the three
member functions live in src/lvba_system.cpp (ROS / OpenCV / Ceres: cannot be built here) -> PARITY UNPINNED for their loops.

Where the reference walks a std::unordered_map<int,int> (image -> observation) the order is unspecified; here, as in the HIP
path, images are visited in the order of their first occurrence in the track's BFS component.  It changes which observations
the greedy view-angle filter keeps only when two candidates tie, and the rounding of sums otherwise.
"""
from __future__ import annotations

import numpy as np

from . import track_oracle as to

f32 = np.float32


def voxel_keys(pw, vox):
    """(int64)(float)(x / vox) with -1 for negatives (src/lvba_system.cpp:1289-1293)."""
    loc = (np.asarray(pw, np.float64) / vox).astype(f32)
    loc = np.where(loc < 0, (loc - f32(1.0)).astype(f32), loc)
    return np.trunc(loc).astype(np.int64)


def render_depth(clouds, scan_poses, scan_times, image_times, Rcw, tcw, intr, width, height, half_w=0.5, vox=0.5):
    """Returns depth [n_images, height, width] float32 (0 = empty)."""
    scan_poses = np.asarray(scan_poses, np.float64).reshape(-1, 12)
    world, frame_keys = [], []
    for c, T in zip(clouds, scan_poses):
        p = np.asarray(c, f32)[:, :3].astype(np.float64)
        pw = p @ T[:9].reshape(3, 3).T + T[9:]
        world.append(pw)
        frame_keys.append(voxel_keys(pw, vox))
    allp = np.concatenate(world) if world else np.zeros((0, 3))
    allk = np.concatenate(frame_keys) if frame_keys else np.zeros((0, 3), np.int64)
    grid = {}
    for i, k in enumerate(map(tuple, allk)):
        grid.setdefault(k, []).append(i)
    per_frame = [set(map(tuple, k)) for k in frame_keys]
    ts = np.asarray(scan_times, np.float64)
    fx, fy, cx, cy, k1, k2, p1, p2 = [float(v) for v in intr]
    out = np.zeros((len(image_times), height, width), f32)
    for m, t_img in enumerate(image_times):
        lo = int(np.searchsorted(ts, t_img - half_w, side="left"))      # std::lower_bound
        hi = int(np.searchsorted(ts, t_img + half_w, side="right"))     # std::upper_bound
        vox_set = set()
        for f in range(lo, hi):
            vox_set |= per_frame[f]
        if not vox_set:
            continue
        idx = np.concatenate([np.asarray(grid[k]) for k in sorted(vox_set)])
        pc = allp[idx] @ np.asarray(Rcw[m], np.float64).reshape(3, 3).T + np.asarray(tcw[m], np.float64)
        Z = pc[:, 2]
        ok = Z >= 1e-3
        pc, Z = pc[ok], Z[ok]
        x, y = pc[:, 0] / Z, pc[:, 1] / Z
        r2 = x * x + y * y
        r4 = r2 * r2
        radial = 1.0 + k1 * r2 + k2 * r4
        xd = x * radial + (2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x))
        yd = y * radial + (p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y)
        uu, vv = fx * xd + cx, fy * yd + cy
        fin = np.isfinite(pc).all(1) & np.isfinite(xd) & np.isfinite(yd) & np.isfinite(uu) & np.isfinite(vv)
        ui = np.trunc(np.where(fin, uu, -1.0)).astype(np.int64)
        vi = np.trunc(np.where(fin, vv, -1.0)).astype(np.int64)
        ins = fin & (ui >= 0) & (ui < width) & (vi >= 0) & (vi < height)
        zf = Z[ins].astype(f32)
        img = np.full(height * width, np.inf, f32)
        np.minimum.at(img, vi[ins] * width + ui[ins], zf)
        img[~np.isfinite(img)] = 0.0
        out[m] = img.reshape(height, width)
    return out


def fetch_depth_bilinear(depth, u, v):
    """utils.hpp:246-275 for a CV_32FC1 image; u, v are float (the keypoint's fields).  Returns d > 0 or None."""
    u, v = f32(u), f32(v)
    h, w = depth.shape
    if u < 0 or v < 0 or u >= w - 1 or v >= h - 1:
        return None
    x, y = int(np.floor(u)), int(np.floor(v))
    du, dv = f32(u - f32(x)), f32(v - f32(y))
    d00, d10, d01, d11 = depth[y, x], depth[y, x + 1], depth[y + 1, x], depth[y + 1, x + 1]
    if d00 <= 0 or d10 <= 0 or d01 <= 0 or d11 <= 0:
        return None
    one = f32(1.0)
    d = f32(f32(f32(f32((one - du) * (one - dv)) * d00) + f32(f32(du * (one - dv)) * d10)) +
            f32(f32((one - du) * dv) * d01)) + f32(f32(du * dv) * d11)
    d = f32(d)
    return d if d > 0 else None


def _view_filter(ids, points, Cw, cos_min):
    """ids: [(img, comp_idx)], points: comp_idx -> 3D point (or one point for all).  Greedy filter of :1052-1080 / :1124-1150."""
    kept, dirs = [], []
    for img, ci in ids:
        d = (points[ci] if isinstance(points, dict) else points) - Cw[img]
        n = np.linalg.norm(d)
        if n < 1e-6:
            continue
        d = d / n
        min_dot = min([float(d @ e) for e in dirs], default=1.0)
        if not dirs or min_dot <= cos_min:
            kept.append((img, ci))
            dirs.append(d)
    return kept


def fuse_track(component_img, component_uv, depth, Rcw, tcw, intr, obser_thr=3, min_view_angle_deg=8.0, reproj_thr=3.0):
    """One BFS component (observations in BFS order; uv float32 keypoint coordinates).
    Returns (status, X, mean_reproj, kept_mask): status 0 dropped, 1 triangulated, 2 depth-fused."""
    n = len(component_img)
    kept_mask = np.zeros(n, np.uint8)
    none = (0, np.zeros(3), np.inf, kept_mask)
    if n < obser_thr:
        return none
    first = {}
    for ci, im in enumerate(component_img):
        first.setdefault(int(im), ci)
    if len(first) < obser_thr:
        return none
    unique = list(first.items())                                             # (img, comp_idx), first-occurrence order
    cos_min = np.cos(min_view_angle_deg * np.pi / 180.0)
    Cw = {im: -np.asarray(Rcw[im]).reshape(3, 3).T @ np.asarray(tcw[im]) for im in first}
    uv64 = np.asarray(component_uv, f32).astype(np.float64)
    R3 = [np.asarray(r, np.float64).reshape(3, 3) for r in Rcw]
    t3 = [np.asarray(t, np.float64) for t in tcw]
    # ---- depth-fused candidate (:1016-1106)
    depth_ok, X_depth, m_depth, kept_depth = False, np.zeros(3), np.inf, []
    if depth is not None:
        pts = {}
        for ci in range(n):
            im = int(component_img[ci])
            u, v = f32(component_uv[ci][0]), f32(component_uv[ci][1])
            d = fetch_depth_bilinear(depth[im], u, v)
            if d is None:
                continue
            xy = to.undistort(intr, float(u), float(v))
            if xy is None:
                continue
            dd = float(d)
            Xc = np.array([xy[0] * dd, xy[1] * dd, dd])
            if not np.all(np.isfinite(Xc)):
                continue
            pts[ci] = R3[im].T @ Xc + (-(R3[im].T @ t3[im]))                 # camToWorld, utils.hpp:277-284
        valid = sorted(pts)
        if len(valid) >= obser_thr:
            anchor = pts[valid[0]]
            inl = [ci for ci in valid if np.linalg.norm(pts[ci] - anchor) < 0.12]
            best = {}
            for ci in inl:
                best.setdefault(int(component_img[ci]), ci)
            if len(best) >= obser_thr:
                X_depth = sum(pts[ci] for ci in best.values()) / float(len(best))
                kd = _view_filter(list(best.items()), pts, Cw, cos_min)
                if len(kd) >= obser_thr:
                    m, cnt = to.mean_reproj(intr, R3, t3, X_depth, [im for im, _ in kd], [uv64[ci] for _, ci in kd], obser_thr)
                    if m is not None:
                        m_depth, kept_depth = m, [ci for _, ci in kd]
                        depth_ok = m <= reproj_thr
    # ---- triangulation candidate (:1108-1160)
    tri_ok, X_tri, m_tri, kept_tri = False, np.zeros(3), np.inf, []
    if len(unique) >= 4:
        ok, Xs, _, _ = to.triangulate_track(intr, R3, t3, [im for im, _ in unique], [uv64[ci] for _, ci in unique])
        if ok:
            kt = _view_filter(unique, Xs, Cw, cos_min)
            kept_tri = [ci for _, ci in kt]
            if len(kt) >= 4:
                ok2, X2, m2, _ = to.triangulate_track(intr, R3, t3, [im for im, _ in kt], [uv64[ci] for _, ci in kt])
                if ok2:
                    X_tri, m_tri = X2, m2
                    tri_ok = m2 <= reproj_thr
    # ---- selection (:1162-1200)
    if depth_ok and tri_ok:
        use_tri = m_tri < m_depth
    elif tri_ok:
        use_tri = True
    elif depth_ok:
        use_tri = False
    else:
        return none
    X, m, kept = (X_tri, m_tri, kept_tri) if use_tri else (X_depth, m_depth, kept_depth)
    if not np.all(np.isfinite(X)) or np.all(np.abs(X) <= 1e-12):             # allFinite / isZero(1e-12)
        return none
    kept_mask[kept] = 1
    return (1 if use_tri else 2), X, m, kept_mask


def fuse_tracks(obs_off, obs_img, obs_uv, depth, Rcw, tcw, intr, **kw):
    n = len(obs_off) - 1
    status = np.zeros(n, np.uint8)
    X = np.zeros((n, 3))
    err = np.full(n, np.inf)
    kept = np.zeros(len(obs_img), np.uint8)
    for t in range(n):
        a, b = int(obs_off[t]), int(obs_off[t + 1])
        s, x, m, k = fuse_track(obs_img[a:b], obs_uv[a:b], depth, Rcw, tcw, intr, **kw)
        status[t], X[t], err[t] = s, x, m
        kept[a:b] = k
    return status, X, err, kept
