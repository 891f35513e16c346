"""One dissected solve under the microscope: synth's parking-lot graph at N poses, `reps` solves, the library's own solve time
(HIP events on its stream).  usage: python tools/nd_probe.py [N] [reps] [voxels_per_pose]"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
vpp = int(sys.argv[3]) if len(sys.argv) > 3 else 100
pkg = importlib.import_module("global-lvba_amd")
synth = importlib.import_module("global-lvba_amd.synth")
d = synth.make_balm_problem(N, vpp * N, revisit="lot", device="cuda")
prob = pkg.BalmProblem(N, d["voxel_off"], d["pose_idx"], d["clusters"])
info = prob.info()
prob.eval(d["poses_init"], want_H=False, want_g=False)
for _ in range(4):
    prob.solve(0.01)
prob.set_profiling(True); prob.profile(reset=True)
for _ in range(reps):
    prob.solve(0.01)
p = prob.profile()
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("LVBA_")}, "nd_kind": info["nd_kind"], "arcs": info["nd_arcs"],
                  "sep": info["nd_sep_poses"], "sep_bb": info["nd_sep_band_blocks"], "solve_ms": p["solve_ms"] / p["solve_calls"],
                  "model": [info["nd_model_band_ms"], info["nd_model_nd_ms"]]}))
