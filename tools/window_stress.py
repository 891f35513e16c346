"""Stress of the threaded window stage: repeated runs at several thread counts must give bitwise identical outputs."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("global-lvba_amd")
synth = importlib.import_module("global-lvba_amd.synth")
s = synth.make_scans(24, 30000, room=(14, 10, 4), n_panels=8, seed=3, rot_sigma_deg=0.1, trans_sigma=0.03)
clouds = s["clouds"] * 8                                  # 192 frames = 48 windows of 4
poses = np.concatenate([np.asarray(s["poses"]).reshape(-1, 12)] * 8)
ref = None
with pkg.Scans(clouds) as scans:
    for rep in range(4):
        for thr in ("1", "3", "8"):
            os.environ["LVBA_WINDOW_THREADS"] = thr
            out = scans.window_ba(poses, window_size=4, voxel_size=1.0, anchor_leaf=0.05)
            pts = out["anchor_scans"].download(5)
            out["anchor_scans"].close()
            key = (out["window_poses"].tobytes(), out["rel_poses"].tobytes(), out["anchor_index"].tobytes(), pts.tobytes())
            if ref is None:
                ref = key
            assert key == ref, (rep, thr)
print("window stress ok: 12 runs x 48 windows identical")
