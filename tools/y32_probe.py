"""LVBA_Y32=1 (fp32 Y records between the factor pass and the pair pass) against the default fp64 records, judged on what
north_star names: per-iteration LM cost and the final poses (<= 1e-5), plus the Hessian blocks and the evaluation time.
usage: python tools/y32_probe.py [C2|C3|NxV ...]   -> one JSON line per configuration"""
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402

pkg = importlib.import_module("global-lvba_amd")
synth = importlib.import_module("global-lvba_amd.synth")


def run(cfg):
    N, V = bench.parse_config(cfg, synth)
    d = synth.make_balm_problem(N, V, device="cuda:0")
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    out = {"config": cfg}
    res = {}
    for mode in ("0", "1"):
        os.environ["LVBA_Y32"] = mode
        prob = pkg.BalmProblem(N, d["voxel_off"], d["pose_idx"], d["clusters"], device=0)
        info = prob.info()
        assert info["y_fp32"] == int(mode), info
        x0 = d["poses_init"]
        bi, bj, blocks, g, c = prob.eval_blocks(x0)
        x, trace, rc = prob.refine(x0)
        prob.set_profiling(True)
        prob.profile(reset=True)
        t0 = time.perf_counter()
        prob.refine(x0)
        dt = time.perf_counter() - t0
        p = prob.profile()
        res[mode] = dict(blocks=blocks, bi=bi, bj=bj, g=g, c=c, x=x, trace=trace, rc=rc,
                         eval_ms=p["eval_ms"] / max(1, p["eval_calls"]), eval_kernel_ms=p["eval_kernel_ms"] / max(1, p["eval_calls"]),
                         ms_per_iter=1e3 * dt / max(1, len(trace)))
        prob.close()
    os.environ["LVBA_Y32"] = "0"
    a, b = res["0"], res["1"]
    assert np.array_equal(a["bi"], b["bi"]) and np.array_equal(a["bj"], b["bj"])
    scale = np.abs(a["blocks"]).reshape(len(a["bi"]), -1).max(1)
    out["H_block_rel_max"] = float((np.abs(a["blocks"] - b["blocks"]).reshape(len(a["bi"]), -1).max(1) / np.maximum(scale, 1e-300)).max())
    out["H_rel_fro"] = float(np.linalg.norm(a["blocks"] - b["blocks"]) / np.linalg.norm(a["blocks"]))
    out["g_equal"] = bool(np.array_equal(a["g"], b["g"]))
    out["cost_equal"] = bool(a["c"] == b["c"])
    n = min(len(a["trace"]), len(b["trace"]))
    out["iterations"] = [len(a["trace"]), len(b["trace"])]
    out["trace_cost_rel_max"] = float(max(max(abs(a["trace"][k]["residual1"] - b["trace"][k]["residual1"]) / abs(a["trace"][k]["residual1"]),
                                              abs(a["trace"][k]["residual2"] - b["trace"][k]["residual2"]) / abs(a["trace"][k]["residual2"]))
                                          for k in range(n)))
    out["accept_pattern_equal"] = [r["accepted"] for r in a["trace"][:n]] == [r["accepted"] for r in b["trace"][:n]]
    out["final_cost_rel"] = float(abs(a["trace"][-1]["residual2"] - b["trace"][-1]["residual2"]) / abs(a["trace"][-1]["residual2"]))
    out["final_pose_abs_max"] = float(np.abs(a["x"] - b["x"]).max())
    out["eval_ms"] = [a["eval_ms"], b["eval_ms"]]
    out["eval_kernel_ms"] = [a["eval_kernel_ms"], b["eval_kernel_ms"]]
    out["ms_per_iter_of_a_refinement"] = [a["ms_per_iter"], b["ms_per_iter"]]
    out["within_north_star_1e-5"] = bool(out["trace_cost_rel_max"] <= 1e-5 and out["final_pose_abs_max"] <= 1e-5)
    return out


if __name__ == "__main__":
    for cfg in (sys.argv[1:] or ["C2", "C3"]):
        print(json.dumps(run(cfg)), flush=True)
