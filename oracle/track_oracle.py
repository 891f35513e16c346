"""CPU oracle of the per-track landmark initialisation.  TEST INFRASTRUCTURE ONLY.

Restates (paths relative to /root/reference)
  undistortPixelToNormalized / distortNormalized / projectWorldToPixel   include/utils.hpp:168-233
  TriangulateTrackDLT                                                     src/lvba_system.cpp:50-111
  ComputeMeanReproj                                                       src/lvba_system.cpp:8-48
The reference walks an unordered_map<image, observation>; the caller passes the observations in that order
(fusion_oracle.umap_order), here it only changes the rounding of the sums.  Each track is the already de-duplicated list (one
observation per image).  PINNED: the camera model (undistort / distort / project) against the reference's own
include/utils.hpp (tests/test_ref_pin.py); the DLT and the mean reprojection error against src/lvba_system.cpp itself, through
BuildTracksAndFuse3D (oracle/ref_glue_system.cpp, tests/test_ref_system.py: landmarks to 1e-14).
"""
from __future__ import annotations

import numpy as np


def undistort(intr, u, v):
    fx, fy, cx, cy, k1, k2, p1, p2 = intr
    if not (np.isfinite(u) and np.isfinite(v)) or abs(fx) < 1e-12 or abs(fy) < 1e-12:
        return None
    xd, yd = (u - cx) / fx, (v - cy) / fy
    xu, yu = xd, yd
    for _ in range(8):
        r2 = xu * xu + yu * yu
        radial = 1.0 + k1 * r2 + k2 * r2 * r2
        if abs(radial) < 1e-12 or not np.isfinite(radial):
            return None
        x_tan = 2.0 * p1 * xu * yu + p2 * (r2 + 2.0 * xu * xu)
        y_tan = p1 * (r2 + 2.0 * yu * yu) + 2.0 * p2 * xu * yu
        xu, yu = (xd - x_tan) / radial, (yd - y_tan) / radial
        if not (np.isfinite(xu) and np.isfinite(yu)):
            return None
    return xu, yu


def project(intr, R, t, X):
    fx, fy, cx, cy, k1, k2, p1, p2 = intr
    Xc = R @ X + t
    if not np.all(np.isfinite(Xc)) or Xc[2] <= 1e-12:
        return None
    x, y = Xc[0] / Xc[2], Xc[1] / Xc[2]
    r2 = x * x + y * y
    radial = 1.0 + k1 * r2 + k2 * r2 * r2
    xd = x * radial + (2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x))          # x * radial + x_tan (utils.hpp:176-179)
    yd = y * radial + (p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y)
    u, v = fx * xd + cx, fy * yd + cy
    return (u, v) if np.isfinite(u) and np.isfinite(v) else None


def mean_reproj(intr, Rcw, tcw, X, cams, uv, min_count):
    s, n = 0.0, 0
    for c, (u, v) in zip(cams, uv):
        p = project(intr, Rcw[c], tcw[c], X)
        if p is None:
            continue
        s += np.hypot(p[0] - u, p[1] - v)
        n += 1
    if n < min_count:
        return None, n
    m = s / n
    return (m if np.isfinite(m) else None), n


def triangulate_track(intr, Rcw, tcw, cams, uv):
    """Returns (ok, X, mean_reproj, count)."""
    if len(cams) < 4:
        return False, np.zeros(3), np.inf, 0
    AtA = np.zeros((4, 4))
    rows = 0
    for c, (u, v) in zip(cams, uv):
        xy = undistort(intr, u, v)
        if xy is None:
            continue
        P = np.hstack([Rcw[c], tcw[c][:, None]])
        ru = xy[0] * P[2] - P[0]
        rv = xy[1] * P[2] - P[1]
        AtA += np.outer(ru, ru) + np.outer(rv, rv)
        rows += 2
    if rows < 8:
        return False, np.zeros(3), np.inf, 0
    w, V = np.linalg.eigh(AtA)
    Xh = V[:, 0]
    if abs(Xh[3]) < 1e-12:
        return False, np.zeros(3), np.inf, 0
    X = Xh[:3] / Xh[3]
    if not np.all(np.isfinite(X)):
        return False, np.zeros(3), np.inf, 0
    m, n = mean_reproj(intr, Rcw, tcw, X, cams, uv, 4)
    return m is not None, X, (m if m is not None else np.inf), n


def triangulate_tracks(intr, Rcw, tcw, obs_off, obs_cam, obs_uv):
    n = len(obs_off) - 1
    ok = np.zeros(n, np.uint8)
    X = np.zeros((n, 3))
    err = np.full(n, np.inf)
    cnt = np.zeros(n, np.int32)
    for i in range(n):
        a, b = obs_off[i], obs_off[i + 1]
        o, x, m, c = triangulate_track(intr, Rcw, tcw, obs_cam[a:b], obs_uv[a:b])
        ok[i], X[i], err[i], cnt[i] = o, x, m, c
    return ok, X, err, cnt
