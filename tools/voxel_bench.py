"""Times lvba_voxmap_build / find_planes on synthetic scans (GPU box).  usage: voxel_bench.py [frames] [pts_per_frame]"""
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("global-lvba_amd")
synth = importlib.import_module("global-lvba_amd.synth")

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ppf = int(sys.argv[2]) if len(sys.argv) > 2 else 250000
t0 = time.time()
s = synth.make_scans(frames, ppf, room=(60, 40, 8), n_panels=40, n_blobs=60, origin=(120.0, -80.0, 2.0), point_floats=12)
gen_s = time.time() - t0
res = dict(frames=frames, pts_per_frame=ppf, gen_s=gen_s)
for vs in (1.0, 0.5):
    best = 1e9
    for rep in range(4):
        t0 = time.time()
        m = pkg.VoxelMap(s["clouds"], s["poses"], vs)
        dt = time.time() - t0
        best = min(best, dt)
        if rep < 3:
            m.close()
    X = np.concatenate([c[::50, :3].astype(np.float64) @ T[:9].reshape(3, 3).T + T[9:] for c, T in zip(s["clouds"], s["poses"])])
    t0 = time.time()
    plane, valid = m.find_planes(X)
    look = time.time() - t0
    t0 = time.time()
    prob = m.tras_opt()
    to_balm = time.time() - t0
    res[f"vs{vs}"] = dict(build_s=best, mpts_per_s=m.info["n_points"] / best / 1e6, info=m.info, lookup_n=len(X),
                          lookup_s=look, hit_frac=float(valid.mean()), to_balm_s=to_balm, balm_info=prob.info())
    prob.close()
    m.close()
print(json.dumps(res, default=int))
