"""Times lvba_depth_render and lvba_fuse_tracks on a synthetic scene (GPU box).  usage: fusion_bench.py [frames] [pts] [tracks]"""
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("global-lvba_amd")
synth = importlib.import_module("global-lvba_amd.synth")
vis = importlib.import_module("global-lvba_amd.visual")

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ppf = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
n_tracks = int(sys.argv[3]) if len(sys.argv) > 3 else 200000
W, H = 1280, 1024
intr = np.array([646.78472, 646.65775, 313.456795 * 2, 261.399612 * 2, -0.076160, 0.123001, -0.00113, 0.000251])
RCB = np.array([[0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]])
s = synth.make_scans(frames, ppf, room=(30, 20, 6), n_panels=0, n_blobs=0, clutter_frac=0.0, rot_sigma_deg=0.0, trans_sigma=0.0)
poses = np.asarray(s["poses_gt"], np.float64).reshape(-1, 12)
times = 100.0 + 0.1 * np.arange(frames)
Rcw = np.array([RCB @ T[:9].reshape(3, 3).T for T in poses])
tcw = np.array([-R @ T[9:] for R, T in zip(Rcw, poses)])
res = dict(frames=frames, pts_per_frame=ppf, images=frames, width=W, height=H)
with pkg.Scans(s["clouds"]) as scans:
    for rep in range(2):
        t0 = time.time()
        d = vis.DepthImages.render(scans, poses, times, times, Rcw, tcw, intr, W, H)
        dt = time.time() - t0
        if rep == 0:
            d.close()
    res["render_s"] = dt
    res["render_ms_per_image"] = 1e3 * dt / frames
    img0 = d.download(frames // 2)
    res["fill"] = float((img0 > 0).mean())
    # tracks: landmarks = map points, observed by the 5 cameras around the scan they came from
    rng = np.random.default_rng(1)
    world = np.concatenate([c[:, :3].astype(np.float64) @ T[:9].reshape(3, 3).T + T[9:] for c, T in zip(s["clouds"], poses)])
    src = rng.integers(0, len(world), n_tracks)
    X = world[src]
    f0 = np.clip(src // ppf - 2, 0, frames - 5)
    off = np.arange(n_tracks + 1, dtype=np.int64) * 5
    img = (f0[:, None] + np.arange(5)[None, :]).astype(np.int32).reshape(-1)
    Xc = np.einsum("oij,oj->oi", Rcw[img], np.repeat(X, 5, axis=0)) + tcw[img]
    z = np.maximum(Xc[:, 2], 1e-3)
    uv = (np.stack([intr[0] * Xc[:, 0] / z + intr[2], intr[1] * Xc[:, 1] / z + intr[3]], 1) + 0.3 * rng.standard_normal((len(img), 2))).astype(np.float32)
    for rep in range(2):
        t0 = time.time()
        st, Xf, err, kept = vis.fuse_tracks(off, img, uv, Rcw, tcw, intr, depth=d)
        dt = time.time() - t0
    res["fuse_s"] = dt
    res["tracks_per_s"] = n_tracks / dt
    res["status_counts"] = np.bincount(st, minlength=3).tolist()
    d.close()
print(json.dumps(res))
