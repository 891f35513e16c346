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
PINNED: the helper functions of include/utils.hpp against the reference's own code (tests/test_ref_pin.py), and the three
member functions against src/lvba_system.cpp itself, compiled unmodified against the stand-ins of oracle/shim
(oracle/ref_glue_system.cpp, tests/test_ref_system.py): depth images bit-identical, tracks identical in membership, kept
observations and order, landmarks to rounding.

Two accidents of the reference are part of its results and are restated as they are:
  * it walks std::unordered_map<int,int> containers (image -> observation: unique_id, best_id, kept_id_*), and the greedy
    view-angle filter depends on that order.  With libstdc++ (GCC 11, the toolchain of this image and of the reference's
    Ubuntu targets) the order is a function of the insertion sequence and of reserve(): `umap_order` below restates it
    (bucket = key % bucket_count; a node goes to the front of its bucket, a new bucket to the front of the list);
  * a component that fails the fusion is released (obs_to_track = -1, :1197) and found again from its next member in scan
    order, i.e. fused again with a different BFS order -- `fuse_components_with_retries`.
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
        big = np.abs(uu) < 1e9
        big &= np.abs(vv) < 1e9                                              # keeps the int cast defined; such pixels are outside anyway
        fin &= big
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


# bucket counts libstdc++'s _Prime_rehash_policy hands out (reserve(n) -> the first entry >= n), read off this image's
# libstdc++ (tests/test_fusion_oracle.py re-derives them from the real container)
UMAP_BUCKETS = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 103, 109, 113, 127,
                137, 139, 149, 157, 167, 179, 193, 199, 211, 227, 241, 257, 277, 293, 313, 337, 359, 383, 409, 439, 467, 503, 541,
                577, 619, 661, 709, 761, 823, 887, 953, 1031, 1109, 1193, 1289, 1381, 1493, 1613, 1741, 1879, 2029, 2179, 2357,
                2549, 2753, 2971, 3209, 3469, 3739, 4027, 4349, 4703, 5087, 5503, 5953, 6427, 6949, 7517, 8123, 8783, 9497, 10273,
                11113, 12011, 12983, 14033, 15173, 16411, 17749, 19183, 20753, 22447, 24281, 26267, 28411, 30727, 33223, 35933,
                38873, 42043, 45481, 49201, 53201, 57557, 62233, 67307, 72817, 78779, 85229, 92203, 99733, 107897, 116731, 126271,
                136607, 147793, 159871, 172933, 187091, 202409)


def umap_bucket_count(reserve_n):
    for b in UMAP_BUCKETS:
        if b >= reserve_n:
            return b
    return UMAP_BUCKETS[-1]


def umap_order(keys, reserve_n):
    """Iteration order (positions into `keys`) of a libstdc++ std::unordered_map<int, ...> that was reserve(reserve_n)'d and then
    received the distinct non-negative `keys` in this order (no rehash: len(keys) <= reserve_n)."""
    B = umap_bucket_count(reserve_n)
    order, first_of_bucket = [], {}                       # order: list of positions; bucket -> its current first position
    for pos, k in enumerate(keys):
        b = int(k) % B
        if b in first_of_bucket:
            order.insert(order.index(first_of_bucket[b]), pos)
        else:
            order.insert(0, pos)
        first_of_bucket[b] = pos
    return order


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
    unique = list(first.items())                                             # (img, comp_idx), insertion order
    unique = [unique[i] for i in umap_order([im for im, _ in unique], n)]    # unique_id.reserve(component.size()), :1005
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
                best = list(best.items())
                best = [best[i] for i in umap_order([im for im, _ in best], len(inl))]   # best_id.reserve(inliers.size())
                X_depth = np.zeros(3)
                for _, ci in best:
                    X_depth = X_depth + pts[ci]
                X_depth = X_depth / float(len(best))
                kd = _view_filter(best, pts, Cw, cos_min)
                if len(kd) >= obser_thr:
                    kdo = [kd[i] for i in umap_order([im for im, _ in kd], len(best))]   # kept_id_depth.reserve(best_id.size())
                    m, cnt = to.mean_reproj(intr, R3, t3, X_depth, [im for im, _ in kdo], [uv64[ci] for _, ci in kdo], obser_thr)
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
                kto = [kt[i] for i in umap_order([im for im, _ in kt], len(unique))]     # kept_id_tri.reserve(unique_id.size())
                ok2, X2, m2, _ = to.triangulate_track(intr, R3, t3, [im for im, _ in kto], [uv64[ci] for _, ci in kto])
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


def build_tracks_and_fuse(keypoints, pairs, matches, depth, Rcw, tcw, intr, obser_thr=3, **kw):
    """The whole loop of BuildTracksAndFuse3D (src/lvba_system.cpp:921-1263): adjacency from the pairwise matches (pairs in
    pairIndex order), BFS components from every still-free key point in (image, key point) scan order, fusion; a component that
    is dropped releases its key points, so it is found again -- in another BFS order -- from its next member (:1000-1014,
    :1197, :1203).  keypoints: list of [n_i, >=2] arrays.  Returns a list of dict(obs [n,2], X, status, err, kept [n])."""
    from collections import deque
    N = len(keypoints)
    nk = [len(k) for k in keypoints]
    adj = [[[] for _ in range(n)] for n in nk]
    order = sorted(range(len(pairs)), key=lambda q: (min(pairs[q]), max(pairs[q])))
    for q in order:
        (i, j), m = pairs[q], np.asarray(matches[q], np.int64).reshape(-1, 2)
        if i > j:
            i, j, m = j, i, m[:, ::-1]
        for ki, kj in m:
            if ki < 0 or kj < 0 or ki >= nk[i] or kj >= nk[j]:
                continue
            adj[i][ki].append((j, int(kj)))
            adj[j][kj].append((i, int(ki)))
    state = [np.full(n, -1, np.int64) for n in nk]
    tracks = []
    for i in range(N):
        for ki in range(nk[i]):
            if state[i][ki] != -1:
                continue
            comp, q = [], deque([(i, ki)])
            state[i][ki] = -2
            while q:
                ci, ck = q.popleft()
                comp.append((ci, ck))
                for ni, nkk in adj[ci][ck]:
                    if state[ni][nkk] == -1:
                        state[ni][nkk] = -2
                        q.append((ni, nkk))
            res = None
            if len(comp) >= obser_thr and len({c for c, _ in comp}) >= obser_thr:
                img = np.array([c for c, _ in comp], np.int32)
                uv = np.array([keypoints[c][k][:2] for c, k in comp], f32).reshape(-1, 2)
                s, X, e, kept = fuse_track(img, uv, depth, Rcw, tcw, intr, obser_thr=obser_thr, **kw)
                if s:
                    res = dict(obs=np.array(comp, np.int32), X=X, status=s, err=e, kept=kept)
            if res is None:
                for ci, ck in comp:
                    state[ci][ck] = -1
                continue
            for ci, ck in comp:
                state[ci][ck] = len(tracks)
            tracks.append(res)
    return tracks
