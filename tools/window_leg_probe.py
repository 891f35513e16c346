"""Stage times of bench.py's front_end.window_ba leg (4 windows x 16 scans x 250 k points): LVBA_TIMING=1 prints them to stderr.
Development tool.  usage: LVBA_TIMING=1 python tools/window_leg_probe.py"""
import importlib
import os
import sys
import time

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
pkg = importlib.import_module("global-lvba_amd")
synth = importlib.import_module("global-lvba_amd.synth")
base = synth.make_scans(8, 250_000, room=(60, 40, 8), n_panels=16, n_blobs=40, origin=(120.0, -80.0, 2.0),
                        rot_sigma_deg=0.1, trans_sigma=0.03, point_floats=12)
clouds, poses = [], []
for r in range(8):
    for c, T in zip(base["clouds"], base["poses"]):
        T = T.copy(); T[9] += 100.0 * r
        clouds.append(c); poses.append(T)
poses = np.asarray(poses)
for k in range(3):
    t0 = time.perf_counter()
    scans = pkg.Scans(clouds)
    print(f"upload {1e3 * (time.perf_counter() - t0):.2f} ms", file=sys.stderr)
    if k < 2:
        scans.close()
best, m = 1e9, None
for _ in range(4):
    if m is not None:
        m.close()
    t0 = time.perf_counter()
    m = scans.voxel_map(poses, 1.0)
    best = min(best, time.perf_counter() - t0)
print(f"voxel map (1.0 m, {m.info['n_points']} points): {1e3 * best:.3f} ms, phases "
      f"{ {k: round(m.info[k], 3) for k in ('key_ms', 'sort_ms', 'count_ms', 'write_ms')} }, {m.info['n_voxels']} plane voxels", file=sys.stderr)
m.close()
for k in range(3):
    print(f"--- window_ba call {k}", file=sys.stderr)
    t0 = time.perf_counter()
    w = scans.window_ba(poses, window_size=16, voxel_size=0.5, anchor_leaf=0.05)
    print(f"window_ba {1e3 * (time.perf_counter() - t0):.2f} ms, {len(w['windows'])} windows, "
          f"{[x['n_iter'] for x in w['windows']]} iterations, anchor points {w['anchor_scans'].n_points if hasattr(w['anchor_scans'], 'n_points') else '?'}",
          file=sys.stderr)
    w["anchor_scans"].close()
