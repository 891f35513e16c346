#!/bin/bash
# C3 with the voxels in RANDOM order (as a hash map would hand them over): evaluation time with the voxels re-laid by their first
# pose at create time (default) and without (LVBA_VOXEL_SORT=0: plain block-major pair lists, scattered gathers)
R=$GRAFT_REPO_ROOT; cd $R
cat > /tmp/shuf.py <<'PY'
import importlib, sys, time, os
import numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
pkg = importlib.import_module("global-lvba_amd"); synth = importlib.import_module("global-lvba_amd.synth")
N, V = synth.CONFIGS["C3"]
d = synth.make_balm_problem(N, V, device="cuda")
off, idx, clu = d["voxel_off"], d["pose_idx"], d["clusters"]
if len(sys.argv) > 1 and sys.argv[1] == "shuffle":
    perm = np.random.default_rng(3).permutation(V)
    k = np.diff(off)
    new_off = np.concatenate([[0], np.cumsum(k[perm])]).astype(np.int64)
    start = off[:-1][perm]
    gather = np.repeat(start - new_off[:-1], k[perm]) + np.arange(new_off[-1])
    off, idx, clu = new_off, idx[gather], clu[gather]
prob = pkg.BalmProblem(N, off, idx, clu)
x = d["poses_init"]
t0 = time.perf_counter(); prob.info(); t1 = time.perf_counter()
prob.set_profiling(True)
for _ in range(3): prob.eval(x, want_H=False, want_g=False)
prob.profile(reset=True)
for _ in range(10): prob.eval(x, want_H=False, want_g=False)
p = prob.profile()
_, _, c = prob.eval(x, want_H=False)
print(sys.argv[1:], os.environ.get("LVBA_VOXEL_SORT", "default"), "setup %.2f s  eval %.3f ms  cost %.12e" % (t1 - t0, p["eval_ms"] / p["eval_calls"], c))
PY
python /tmp/shuf.py sorted
python /tmp/shuf.py shuffle
LVBA_VOXEL_SORT=0 python /tmp/shuf.py shuffle
