"""Visual stage alone at the C3 size (2 000 cameras x 125 000 landmarks x 500 k observations): wall time of whole refinements
with different iteration caps, so that the cost of ONE LM iteration inside the loop (no uploads, no first evaluation, no
download) is the slope.  usage: python tools/visual_bench.py [n_cams] [reps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import lvba_amd as pkg  # noqa: E402
from lvba_amd import synth  # noqa: E402


def main():
    n_cams = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    d = synth.make_visual_problem(n_cams, 125_000 * n_cams // 2000, rot_sigma_deg=0.3, trans_sigma=0.10, point_sigma=0.30, device="cuda:0")
    prob = pkg.VisualProblem(n_cams, d["obs_off"], d["obs_cam"], d["obs_uv"], d["plane"], d["valid"], d["intr"], device=0)
    prob.refine(d["q"], d["t"], d["X"], max_iter=3)
    prob.refine(d["q"], d["t"], d["X"], max_iter=3)
    out = {}
    for cap in (2, 4, 6, 50):
        best, its = 1e9, 0
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, trace, term, rc = prob.refine(d["q"], d["t"], d["X"], max_iter=cap)
            best = min(best, time.perf_counter() - t0)
            its = len(trace) - 1
        out[f"cap_{cap}"] = {"iterations": its, "ms": 1e3 * best, "termination": term}
    a, b = out["cap_2"], out["cap_50"]
    if b["iterations"] > a["iterations"]:
        out["ms_per_iteration_in_loop"] = (b["ms"] - a["ms"]) / (b["iterations"] - a["iterations"])
        out["fixed_ms"] = a["ms"] - a["iterations"] * out["ms_per_iteration_in_loop"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
