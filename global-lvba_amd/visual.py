"""Host-side mirror of the reference's visual stage over the C-ABI.

`VisualProblem` wraps an `lvba_visual_t` handle.  `optimize_camera_poses` mirrors the solve inside
`LvbaSystem::optimizeCameraPoses` (reference src/lvba_system.cpp:1509-1665): it takes what that function has at
line 1571 -- camera rotations/translations (T_cam<-world), fused landmarks, per-track inlier observations, the
per-landmark local planes -- and returns the refined cameras and landmarks.  All arithmetic runs in
liblvba_hip.so on the GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


class VisualProblem:
    def __init__(self, n_cams, obs_off, obs_cam, obs_uv, plane, valid, intr, sigma_px=0.5, sigma_plane=0.01, device=0):
        self.lib = L.load()
        self.n_cams = int(n_cams)
        obs_off = np.ascontiguousarray(obs_off, np.int64)
        self.n_tracks = len(obs_off) - 1
        obs_cam = np.ascontiguousarray(obs_cam, np.int32)
        obs_uv = np.ascontiguousarray(obs_uv, np.float64).reshape(-1, 2)
        plane = np.ascontiguousarray(plane, np.float64).reshape(-1, 4)
        valid = np.ascontiguousarray(valid, np.uint8)
        intr = np.ascontiguousarray(intr, np.float64)
        if plane.shape[0] != self.n_tracks or valid.shape[0] != self.n_tracks or intr.shape[0] != 8:
            raise ValueError("plane/valid must have one row per track; intr = fx fy cx cy k1 k2 p1 p2")
        self._h = C.c_void_p()
        L.check(self.lib.lvba_visual_create(self.n_cams, self.n_tracks, obs_off, obs_cam.ctypes.data, obs_uv.ctypes.data,
                                            plane.reshape(-1), valid, intr, float(sigma_px), float(sigma_plane), int(device),
                                            C.byref(self._h)))
        self._keep = (obs_cam, obs_uv)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.lvba_visual_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _state(self, q, t, X):
        q = np.ascontiguousarray(q, np.float64).reshape(-1)
        t = np.ascontiguousarray(t, np.float64).reshape(-1)
        X = np.ascontiguousarray(X, np.float64).reshape(-1)
        if q.size != 4 * self.n_cams or t.size != 3 * self.n_cams or X.size != 3 * self.n_tracks:
            raise ValueError("q [M,4], t [M,3], X [T,3] expected")
        return q, t, X

    def cost(self, q, t, X):
        out = C.c_double()
        L.check(self.lib.lvba_visual_cost(self._h, *self._state(q, t, X), C.byref(out)))
        return out.value

    def linearize(self, q, t, X, radius=1e4):
        n = 6 * self.n_cams
        S, rhs, c = np.empty((n, n)), np.empty(n), C.c_double()
        L.check(self.lib.lvba_visual_linearize(self._h, *self._state(q, t, X), float(radius), S.ctypes.data, rhs.ctypes.data,
                                               C.byref(c)))
        return S, rhs, c.value

    def dist_init(self, n_ranks, rank, uid):
        """Track shards over several ranks (lvba_visual_dist_init): this handle holds the rank's own tracks."""
        L.check(self.lib.lvba_visual_dist_init(self._h, int(n_ranks), int(rank), bytes(uid)))

    def dist_init_external(self, n_ranks, rank, fn, ctx):
        """The caller's all-reduce instead of RCCL (lvba_visual_dist_init_external)."""
        L.check(self.lib.lvba_visual_dist_init_external(self._h, int(n_ranks), int(rank), C.c_void_p(fn), C.c_void_p(ctx)))

    def linearize_only(self, q, t, X, radius=1e4):
        """The factor kernels of one linearisation (residuals, Jacobians, column norms, Schur products -> reduced system on
        the device) without exporting S / rhs; returns the cost."""
        c = C.c_double()
        L.check(self.lib.lvba_visual_linearize(self._h, *self._state(q, t, X), float(radius), None, None, C.byref(c)))
        return c.value

    def info(self):
        i = L.BalmInfo()
        L.check(self.lib.lvba_visual_info(self._h, C.byref(i)))
        return {f: getattr(i, f) for f, _ in i._fields_}

    @staticmethod
    def default_opts(**kw):
        o = L.VisualOpts()
        L.load().lvba_visual_default_opts(C.byref(o))
        for k, v in kw.items():
            setattr(o, k, v)
        return o

    def refine(self, q, t, X, **opts):
        o = self.default_opts(**opts)
        q, t, X = (a.copy() for a in self._state(q, t, X))
        cap = o.max_iter + 2
        trace = (L.VisualTrace * cap)()
        nt, term = C.c_int32(), C.c_int32()
        rc = L.check(self.lib.lvba_visual_refine(self._h, q, t, X, C.byref(o), trace, cap, C.byref(nt), C.byref(term)),
                     allow_numeric=True)
        return (q.reshape(-1, 4), t.reshape(-1, 3), X.reshape(-1, 3)), [trace[i].as_dict() for i in range(nt.value)], \
            L.TERMINATION.get(term.value, str(term.value)), rc


def optimize_camera_poses(qs, ts, Xs, obs_off, obs_cam, obs_uv, plane_n, plane_d, intr, sigma_px=0.5, sigma_plane=0.01,
                          device=0):
    """The ceres::Problem / ceres::Solve region of LvbaSystem::optimizeCameraPoses (src/lvba_system.cpp:1571-1665).
    A landmark has a valid plane iff its normal is finite and non-zero (has_valid_plane, :1596)."""
    plane_n = np.asarray(plane_n, np.float64).reshape(-1, 3)
    plane_d = np.asarray(plane_d, np.float64).reshape(-1)
    valid = (np.isfinite(plane_n).all(1) & np.isfinite(plane_d) & ~(np.abs(plane_n) <= 1e-6).all(1)).astype(np.uint8)
    prob = VisualProblem(len(qs), obs_off, obs_cam, obs_uv, np.concatenate([plane_n, plane_d[:, None]], 1), valid, intr,
                         sigma_px, sigma_plane, device)
    try:
        return prob.refine(qs, ts, Xs) + (valid,)
    finally:
        prob.close()


def triangulate_tracks(Rcw, tcw, obs_off, obs_cam, obs_uv, intr, device=0):
    """TriangulateTrackDLT + ComputeMeanReproj for every track (src/lvba_system.cpp:8-111).
    Returns (ok [n] uint8, X [n,3], mean_reproj [n], count [n])."""
    lib = L.load()
    Rcw = np.ascontiguousarray(Rcw, np.float64).reshape(-1, 9)
    tcw = np.ascontiguousarray(tcw, np.float64).reshape(-1, 3)
    obs_off = np.ascontiguousarray(obs_off, np.int64)
    obs_cam = np.ascontiguousarray(obs_cam, np.int32)
    obs_uv = np.ascontiguousarray(obs_uv, np.float64).reshape(-1, 2)
    n = len(obs_off) - 1
    X = np.zeros((max(n, 1), 3))
    err = np.zeros(max(n, 1))
    cnt = np.zeros(max(n, 1), np.int32)
    ok = np.zeros(max(n, 1), np.uint8)
    L.check(lib.lvba_triangulate_tracks(int(device), len(Rcw), n, obs_off, obs_cam.ctypes.data, obs_uv.ctypes.data,
                                        Rcw.reshape(-1), tcw.reshape(-1), np.ascontiguousarray(intr, np.float64), X.reshape(-1),
                                        err, cnt, ok))
    return ok[:n], X[:n], err[:n], cnt[:n]


def build_tracks(n_keypoints, pairs, matches, obser_thr=3):
    """Feature tracks = connected components of the match graph, as LvbaSystem::BuildTracksAndFuse3D builds them
    (src/lvba_system.cpp:923-1003): adjacency filled pair by pair (i < j) in the given order, BFS from every unvisited
    (image, keypoint) in image-major order, components smaller than obser_thr or seen by fewer than obser_thr images
    dropped, then one observation per image (the first the BFS met).  Host-side graph walk (this is control flow, not a
    kernel).  n_keypoints[i] = number of keypoints of image i; pairs = [(i, j)], matches[k] = [m, 2] indices for pairs[k].
    Returns (obs_off [T+1], obs_img [O], obs_kp [O]) of the de-duplicated tracks."""
    from collections import deque
    N = len(n_keypoints)
    adj = [dict() for _ in range(N)]                    # image -> {kp: [(image, kp), ...]}
    for (i, j), m in zip(pairs, matches):
        if i > j:
            i, j, m = j, i, np.asarray(m)[:, ::-1]
        for ki, kj in np.asarray(m, np.int64).reshape(-1, 2):
            if ki < 0 or kj < 0 or ki >= n_keypoints[i] or kj >= n_keypoints[j]:
                continue
            adj[i].setdefault(int(ki), []).append((j, int(kj)))
            adj[j].setdefault(int(kj), []).append((i, int(ki)))
    seen = [set() for _ in range(N)]
    off, img, kp = [0], [], []
    for i in range(N):
        for ki in sorted(adj[i]):                       # keypoints without matches are singletons: dropped anyway
            if ki in seen[i]:
                continue
            comp, q = [], deque([(i, ki)])
            seen[i].add(ki)
            while q:
                ci, ck = q.popleft()
                comp.append((ci, ck))
                for ni, nk in adj[ci].get(ck, ()):
                    if nk not in seen[ni]:
                        seen[ni].add(nk)
                        q.append((ni, nk))
            if len(comp) < obser_thr:
                for ci, ck in comp:
                    seen[ci].discard(ck)                # upstream resets obs_to_track to -1 (:980)
                continue
            uniq = {}
            for ci, ck in comp:
                uniq.setdefault(ci, ck)
            if len(uniq) < obser_thr:
                for ci, ck in comp:
                    seen[ci].discard(ck)
                continue
            for ci, ck in uniq.items():
                img.append(ci); kp.append(ck)
            off.append(len(img))
    return np.asarray(off, np.int64), np.asarray(img, np.int32), np.asarray(kp, np.int32)


def triangulate_and_filter(Rcw, tcw, keypoints, obs_off, obs_img, obs_kp, intr, min_view_angle_deg=8.0,
                           reproj_mean_thr_px=3.0, device=0):
    """The triangulation candidate of BuildTracksAndFuse3D alone (src/lvba_system.cpp:1108-1160): seed DLT over all images of
    the track, greedy view-angle filter against the seed, DLT again over the kept observations, accepted if its mean
    reprojection error <= reproj_mean_thr_px -- i.e. fuse_tracks() without depth images (one kernel; the images are visited in
    the order of the reference's unordered_map, see csrc/tracks_device.h).  Tracks as build_tracks returns them.
    Returns (ok [T], X [T,3], mean_reproj [T], kept_off [T+1], kept_img, kept_kp)."""
    obs_off = np.asarray(obs_off, np.int64)
    obs_img = np.asarray(obs_img, np.int32)
    obs_kp = np.asarray(obs_kp, np.int32)
    uv = np.array([keypoints[i][k][:2] for i, k in zip(obs_img, obs_kp)], np.float32).reshape(-1, 2)
    status, X, err, kept = fuse_tracks(obs_off, obs_img, uv, Rcw, tcw, intr, depth=None, obser_thr=3,
                                       min_view_angle_deg=min_view_angle_deg, reproj_mean_thr_px=reproj_mean_thr_px, device=device)
    sel = np.nonzero(kept)[0]
    track_of = np.repeat(np.arange(len(obs_off) - 1), np.diff(obs_off))
    koff = np.concatenate([[0], np.cumsum(np.bincount(track_of[sel], minlength=len(obs_off) - 1))]).astype(np.int64)
    return status == 1, X, err, koff, obs_img[sel], obs_kp[sel]


class DepthImages:
    """Device-resident depth images (lvba_depth_t): rendered from the refined LiDAR map
    (LvbaSystem::buildGridMapFromOptimized + generateDepthWithVoxel, src/lvba_system.cpp:835-919, 1266-1338) or uploaded."""

    def __init__(self, handle):
        self._h = handle
        self.lib = L.load()
        n, w, h = C.c_int32(), C.c_int32(), C.c_int32()
        L.check(self.lib.lvba_depth_info(self._h, C.byref(n), C.byref(w), C.byref(h)))
        self.n_images, self.width, self.height = n.value, w.value, h.value

    @classmethod
    def render(cls, scans, scan_poses, scan_times, image_times, Rcw, tcw, intr, width, height, half_window_s=0.5,
               voxel_size=0.5):
        """scans: a voxel.Scans; scan_poses [n,12] (refined, T_world<-body); scan_times ascending; image_times are rounded
        to 1e-6 s as the reference's std::to_string round trip does (src/lvba_system.cpp:1311-1318)."""
        lib = L.load()
        img_t = np.array([float(f"{t:.6f}") for t in np.asarray(image_times, np.float64)], np.float64)
        h = C.c_void_p()
        L.check(lib.lvba_depth_render(scans._h, np.ascontiguousarray(scan_poses, np.float64).reshape(-1),
                                      np.ascontiguousarray(scan_times, np.float64), len(img_t), img_t,
                                      np.ascontiguousarray(Rcw, np.float64).reshape(-1), np.ascontiguousarray(tcw, np.float64).reshape(-1),
                                      np.ascontiguousarray(intr, np.float64), int(width), int(height), float(half_window_s),
                                      float(voxel_size), C.byref(h)))
        return cls(h)

    @classmethod
    def upload(cls, depth, device=0):
        depth = np.ascontiguousarray(depth, np.float32)
        n, hh, ww = depth.shape
        h = C.c_void_p()
        L.check(L.load().lvba_depth_upload(int(device), n, ww, hh, depth.reshape(-1), C.byref(h)))
        return cls(h)

    def download(self, image):
        out = np.zeros(self.width * self.height, np.float32)
        L.check(self.lib.lvba_depth_download(self._h, int(image), out))
        return out.reshape(self.height, self.width)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.lvba_depth_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def fuse_tracks(obs_off, obs_img, obs_uv, Rcw, tcw, intr, depth=None, obser_thr=3, min_view_angle_deg=8.0,
                reproj_mean_thr_px=3.0, device=0):
    """The per-component part of LvbaSystem::BuildTracksAndFuse3D (src/lvba_system.cpp:1000-1225) on the GPU: depth-fused and
    triangulated candidates + selection for every track (BFS component, observations in BFS order, float keypoints).
    depth: a DepthImages or None.  Returns (status [n] 0 dropped / 1 triangulated / 2 depth-fused, X [n,3], mean_reproj [n],
    kept [O] mask of the inlier observations)."""
    lib = L.load()
    Rcw = np.ascontiguousarray(Rcw, np.float64).reshape(-1, 9)
    tcw = np.ascontiguousarray(tcw, np.float64).reshape(-1, 3)
    obs_off = np.ascontiguousarray(obs_off, np.int64)
    obs_img = np.ascontiguousarray(obs_img, np.int32)
    obs_uv = np.ascontiguousarray(obs_uv, np.float32).reshape(-1, 2)
    n, O = len(obs_off) - 1, len(obs_img)
    status = np.zeros(max(n, 1), np.uint8)
    X = np.zeros((max(n, 1), 3))
    err = np.zeros(max(n, 1))
    kept = np.zeros(max(O, 1), np.uint8)
    o = L.FuseOpts()
    lib.lvba_fuse_default_opts(C.byref(o))
    o.obser_thr, o.min_view_angle_deg, o.reproj_mean_thr_px = int(obser_thr), float(min_view_angle_deg), float(reproj_mean_thr_px)
    L.check(lib.lvba_fuse_tracks(int(device), depth._h if depth is not None else None, len(Rcw), Rcw.reshape(-1), tcw.reshape(-1),
                                 np.ascontiguousarray(intr, np.float64), n, obs_off, obs_img.ctypes.data, obs_uv.ctypes.data,
                                 C.byref(o), status.ctypes.data, X.reshape(-1), err, kept.ctypes.data))
    return status[:n], X[:n], err[:n], kept[:O]
