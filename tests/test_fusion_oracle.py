"""CPU tests of oracle/fusion_oracle.py (depth rendering + per-track fusion): hand-built cases for every rule of
LvbaSystem::generateDepthWithVoxel / BuildTracksAndFuse3D, and fetchDepthBilinear pinned against the reference's own
include/utils.hpp where oracle/_ref is available."""
import numpy as np
import pytest

import oracle
from oracle import fusion_oracle as fo

RCB = np.array([[0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]])      # camera z = body x
INTR = np.array([120.0, 118.0, 80.0, 60.0, 0.02, -0.005, 0.001, -0.0005])


def _cam(Rwb, pwb):
    Rcw = RCB @ Rwb.T
    return Rcw, -Rcw @ pwb


def test_depth_is_a_float_z_buffer_with_time_window_and_voxel_union():
    # a wall at x = 5 seen by scan 0 (t = 0.0) and a nearer panel at x = 2 seen by scan 1 (t = 10.0), same lines of sight
    yy, zz = np.meshgrid(np.linspace(-1, 1, 60), np.linspace(-0.7, 0.7, 40))
    wall = np.stack([np.full(yy.size, 5.0), yy.ravel() * 2.5, zz.ravel() * 2.5], 1).astype(np.float32)
    panel = np.stack([np.full(yy.size, 2.0), yy.ravel(), zz.ravel()], 1).astype(np.float32)
    I12 = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], float)
    Rcw, tcw = _cam(np.eye(3), np.zeros(3))
    args = dict(Rcw=[Rcw, Rcw], tcw=[tcw, tcw], intr=INTR, width=160, height=120)
    d = fo.render_depth([wall, panel], [I12, I12], [0.0, 10.0], [0.2, 10.2], **args)
    assert d.dtype == np.float32 and d.shape == (2, 120, 160)
    # image 0 only sees the voxels scan 0 touched (the wall); image 1 only those of scan 1 (the panel)
    assert np.isclose(d[0][d[0] > 0], 5.0).all() and np.isclose(d[1][d[1] > 0], 2.0).all()
    assert (d[0] > 0).sum() > 1000 and (d[1] > 0).sum() > 1000
    # one image whose window covers both scans: the nearer surface wins where both project
    d2 = fo.render_depth([wall, panel], [I12, I12], [0.0, 0.4], [0.2], Rcw=[Rcw], tcw=[tcw], intr=INTR, width=160, height=120)
    both = (d[0] > 0) & (d[1] > 0)
    assert both.sum() > 500 and np.isclose(d2[0][both], 2.0).all()
    # a point in a voxel the window's scans touched is drawn even if it came from a scan OUTSIDE the window (grid_map_ is global)
    far = np.array([[5.0, 0.0, 0.0]], np.float32)                          # same voxel as the wall centre
    d3 = fo.render_depth([wall, far], [I12, I12], [0.0, 99.0], [0.2], Rcw=[Rcw], tcw=[tcw], intr=INTR, width=160, height=120)
    assert (d3[0] > 0).sum() >= (d[0] > 0).sum()
    # points behind the camera or closer than 1 mm are skipped
    d4 = fo.render_depth([np.array([[-3.0, 0, 0], [5e-4, 0, 0]], np.float32)], [I12], [0.0], [0.0], Rcw=[Rcw], tcw=[tcw],
                         intr=INTR, width=160, height=120)
    assert not d4.any()


def test_fetch_depth_bilinear_matches_reference_and_rules():
    rng = np.random.default_rng(3)
    depth = (2.0 + rng.random((40, 50))).astype(np.float32)
    depth[10, 10] = 0.0
    assert fo.fetch_depth_bilinear(depth, 9.5, 9.5) is None                 # one of the four neighbours is empty
    assert fo.fetch_depth_bilinear(depth, -0.1, 5) is None and fo.fetch_depth_bilinear(depth, 49.0, 5) is None   # u >= w-1
    assert fo.fetch_depth_bilinear(depth, 48.99, 38.99) is not None
    if oracle.Reference.available():
        ref = oracle.Reference()
        for _ in range(300):
            u, v = np.float32(rng.uniform(-1, 50)), np.float32(rng.uniform(-1, 40))
            a, b = fo.fetch_depth_bilinear(depth, u, v), ref.fetch_depth_bilinear(depth, u, v)
            assert (a is None) == (b is None)
            if a is not None:
                assert abs(float(a) - b) <= 4e-7 * b                        # float arithmetic; the compiler may contract to FMAs


def _scene():
    rng = np.random.default_rng(5)
    n_img = 6
    Rs, ts = [], []
    for m in range(n_img):                                                   # cameras on a line, looking along +x
        Rcw, tcw = _cam(np.eye(3), np.array([0.0, 0.5 * m - 1.2, 0.05 * m]))
        Rs.append(Rcw); ts.append(tcw)
    # landmarks far enough apart in every image that the 6 x 6 depth patches painted around them never overlap
    yv = np.linspace(-1.9, 2.9, 30)
    X = np.stack([np.full(30, 6.0) + 0.5 * np.sin(np.arange(30)), yv, np.where(np.arange(30) % 3 == 0, -0.8, np.where(np.arange(30) % 3 == 1, 0.0, 0.8))], 1)
    return n_img, Rs, ts, X, rng


def _observe(n_img, Rs, ts, X, rng, noise=0.3):
    off, img, uv = [0], [], []
    from oracle import track_oracle as to
    for x in X:
        for m in range(n_img):
            p = to.project(INTR, Rs[m], ts[m], x)
            if p is not None and 1 < p[0] < 158 and 1 < p[1] < 118:
                img.append(m); uv.append(np.float32(p) + np.float32(noise * rng.standard_normal(2)))
        off.append(len(img))
    return np.array(off), np.array(img, np.int32), np.array(uv, np.float32)


def test_triangulation_candidate_alone():
    n_img, Rs, ts, X, rng = _scene()
    off, img, uv = _observe(n_img, Rs, ts, X, rng)
    st, Xf, err, kept = fo.fuse_tracks(off, img, uv, None, Rs, ts, INTR)
    assert (st == 1).sum() >= 25 and not (st == 2).any()
    good = st == 1
    assert np.abs(Xf[good] - X[good]).max() < 0.6 and (err[good] <= 3.0).all()
    for t in np.nonzero(good)[0]:
        assert kept[off[t]:off[t + 1]].sum() >= 4                           # second DLT needs >= 4 kept observations
    # fewer than obser_thr observations / images: dropped
    st2, _, _, _ = fo.fuse_tracks(np.array([0, 2]), img[:2], uv[:2], None, Rs, ts, INTR)
    assert st2[0] == 0
    # duplicates of one image do not count as images
    st3, _, _, _ = fo.fuse_tracks(np.array([0, 4]), np.array([0, 0, 0, 1], np.int32), uv[:4], None, Rs, ts, INTR)
    assert st3[0] == 0


def test_depth_candidate_and_selection():
    n_img, Rs, ts, X, rng = _scene()
    off, img, uv = _observe(n_img, Rs, ts, X, rng, noise=0.05)
    # perfect depth images: every pixel holds the depth of the landmark nearest in the image (piecewise constant is enough
    # for bilinear sampling to return that depth around the keypoint)
    depth = np.zeros((n_img, 120, 160), np.float32)
    for m in range(n_img):
        for x in X:
            Xc = Rs[m] @ x + ts[m]
            from oracle import track_oracle as to
            p = to.project(INTR, Rs[m], ts[m], x)
            if p is None:
                continue
            u, v = int(p[0]), int(p[1])
            depth[m, max(0, v - 2):v + 4, max(0, u - 2):u + 4] = np.float32(Xc[2])
    st, Xf, err, kept = fo.fuse_tracks(off, img, uv, depth, Rs, ts, INTR)
    assert (st > 0).sum() >= 25 and (st == 2).sum() >= 1 and (st == 1).sum() >= 1   # both candidates win somewhere
    ok = st > 0
    assert np.abs(Xf[ok] - X[ok]).max() < 0.25
    # with depth only for THREE images a track can still be depth-fused (obser_thr = 3) where triangulation needs 4
    # (images 0, 2, 5: an observation is kept when its ray differs by >= 8 degrees from at least one ray already kept -- at 8 m
    # that needs > 1.1 m of baseline to a kept camera; the rays are visited in the order of the reference's unordered_map, for
    # these keys 5, 2, 0.  Images 0, 3, 5 are visited 5, 3, 0 (3 and 0 share a bucket) and lose image 3 to image 5.)
    sel = [o for o in range(off[0], off[1]) if img[o] in (0, 2, 5)]
    assert len(sel) == 3
    s3, X3, e3, k3 = fo.fuse_tracks(np.array([0, 3]), img[sel], uv[sel], depth, Rs, ts, INTR)
    assert s3[0] == 2 and k3.sum() == 3 and np.abs(X3[0] - X[0]).max() < 0.25
    sel_b = [o for o in range(off[0], off[1]) if img[o] in (0, 3, 5)]
    assert fo.fuse_tracks(np.array([0, 3]), img[sel_b], uv[sel_b], depth, Rs, ts, INTR)[0][0] == 0
    sel2 = [o for o in range(off[0], off[1]) if img[o] in (0, 1, 2)]         # all within 1 m of camera 0: only one ray survives
    s3b, _, _, _ = fo.fuse_tracks(np.array([0, 3]), img[sel2], uv[sel2], depth, Rs, ts, INTR)
    assert s3b[0] == 0
    # a wrong depth in one image breaks the 0.12 m consistency with the first valid observation: that observation is left out
    bad = depth.copy()
    bad[int(img[off[1] + 2])] += 1.0
    s4, X4, e4, k4 = fo.fuse_tracks(off[1:3] - off[1], img[off[1]:off[2]], uv[off[1]:off[2]], bad, Rs, ts, INTR)
    if s4[0] == 2:
        assert k4[2] == 0
