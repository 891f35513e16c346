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
