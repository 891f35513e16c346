"""Wall time per LM step of the headline problem WITHOUT the library's event profiling (bench.py keeps it on for the stage
times): A/B of host-side changes to the LM driver.  usage: python tools/lm_steps_bench.py [steps] [config]"""
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

pkg = importlib.import_module("global-lvba_amd")
synth = importlib.import_module("global-lvba_amd.synth")


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    N, V = bench.parse_config(sys.argv[2] if len(sys.argv) > 2 else "C3", synth)
    d = synth.make_balm_problem(N, V, device="cuda:0")
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    prob = pkg.BalmProblem(N, d["voxel_off"], d["pose_idx"], d["clusters"], device=0)
    x0 = d["poses_init"]
    prob.refine(x0)
    active = False
    rows = []

    def step():
        nonlocal active
        if not active:
            prob.lm_begin(x0)
            active = True
        row, done, rc = prob.lm_step()
        rows.append((row["accepted"], row["residual2"]))
        if done or rc != 0:
            prob.lm_end(want_poses=False)
            active = False

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"steps": steps, "ms_per_step": 1e3 * dt / steps, "accepted": sum(a for a, _ in rows[5:]), "last_cost": rows[-1][1]}))


if __name__ == "__main__":
    main()
