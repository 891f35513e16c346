"""Envelope (profile) of the pose-block Hessian under the solver's ordering vs the full band: how many 64x64 tiles of the
trailing update are structurally zero.  usage: envelope_probe.py [config]"""
import importlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("global-lvba_amd")
synth = importlib.import_module("global-lvba_amd.synth")
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
N, V = bench.parse_config(cfg, synth)
d = synth.make_balm_problem(N, V, device="cuda:0")
to_np = lambda a: a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)
voff, pidx, clu = to_np(d["voxel_off"]), to_np(d["pose_idx"]), d["clusters"]
c = dict(n_poses=N)
prob = pkg.BalmProblem(c["n_poses"], voff, pidx, clu)
info = prob.info()
perm = prob.ordering()
N = c["n_poses"]
iperm = np.empty(N, np.int64); iperm[perm] = np.arange(N)
rank = iperm[pidx]
vmin = np.minimum.reduceat(rank, voff[:-1])
vox_of = np.repeat(np.arange(len(voff) - 1), np.diff(voff))
first = np.arange(N)
np.minimum.at(first, rank, vmin[vox_of])
Bb = info["band_blocks"]
# envelope closure under elimination: first is already the row-wise minimum column; fill stays inside [first[r], r]
prof = int((np.arange(N) - first).sum())
band = int(np.minimum(np.arange(N), Bb).sum())
# per 64-column panel: row tiles (64 rows = 64/6 poses... work in scalar rows) that hold a non-zero of the panel
n = 6 * N
first_row = np.repeat(6 * first, 6)                       # first non-zero scalar column of every scalar row
bw = 6 * Bb + 5
tiles_full = tiles_env = 0
for k in range(0, n, 64):
    w0, rend = k + 64, min(n, k + 64 + bw)
    if w0 >= rend:
        continue
    T = (rend - w0 + 63) // 64
    act = np.zeros(T, bool)
    rows = np.arange(w0, rend)
    nz = first_row[rows] < k + 64                           # the row reaches into this panel's columns
    np.logical_or.at(act, (rows - w0) // 64, nz)
    a = int(act.sum())
    tiles_full += T * (T + 1) // 2
    tiles_env += a * (a + 1) // 2
print(json.dumps(dict(config=cfg, band_blocks=int(Bb), profile_blocks=prof, band_slots=band, fill=prof / band,
                      update_tiles_full=tiles_full, update_tiles_in_envelope=tiles_env, ratio=tiles_env / tiles_full)))
