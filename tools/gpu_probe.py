"""GPU probe: generate a BASELINE config on the GPU, run each stage of the hot path with profiling and
print per-stage times / achieved bandwidth.  Usage: python tools/gpu_probe.py C2 [C3] [--lm]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    pkg = importlib.import_module("global-lvba_amd")
    synth = importlib.import_module("global-lvba_amd.synth")
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["C2"]
    for name in names:
        if "x" in name:
            N, V = (int(t) for t in name.split("x"))
        else:
            N, V = synth.CONFIGS[name]
        t0 = time.time()
        import torch
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        d = synth.make_balm_problem(N, V, device=dev)
        if dev == "cuda":
            torch.cuda.synchronize(); torch.cuda.empty_cache()
        F = len(d["pose_idx"])
        print(f"[{name}] N={N} V={V} F={F} generated on {dev} in {time.time()-t0:.1f}s", flush=True)
        t0 = time.time()
        for tag, kw in (("band", {}),) + ((("dense", dict(band_frac=0.0)),) if "--dense" in sys.argv else ()):
            prob = pkg.BalmProblem(N, d["voxel_off"], d["pose_idx"], d["clusters"], **kw)
            info = prob.info()
            print(f"[{name}/{tag}] create+finalize {time.time()-t0:.1f}s info={info}", flush=True)
            prob.set_profiling(True)
            x = d["poses_init"]
            c = prob.cost(x, True)
            prob.profile(reset=True)
            for _ in range(10):
                prob.cost(x, True)
            p = prob.profile(reset=True)
            bytes_cost = 84 * F + 4 * (V + 1) + 96 * N + 8
            kms = p["cost_kernel_ms"] / p["cost_calls"]
            print(f"[{name}/{tag}] cost={c:.6e} cost stage {p['cost_ms']/p['cost_calls']:.3f} ms, kernel {kms:.3f} ms "
                  f"-> {bytes_cost/kms/1e6:.1f} GB/s algorithmic", flush=True)
            prob.eval(x, want_H=False, want_g=False)
            prob.profile(reset=True)
            for _ in range(5):
                prob.eval(x, want_H=False, want_g=False)
            p = prob.profile(reset=True)
            print(f"[{name}/{tag}] eval stage {p['eval_ms']/p['eval_calls']:.3f} ms, kernel {p['eval_kernel_ms']/p['eval_calls']:.3f} ms "
                  f"(Q={info['n_pairs']} pairs, {info['n_pairs']*36+F*27} atomics)", flush=True)
            t1 = time.time()
            dx = prob.solve(0.01)
            p = prob.profile(reset=True)
            print(f"[{name}/{tag}] solve first {p['solve_ms']:.3f} ms (wall {1e3*(time.time()-t1):.1f} ms) |dx|max={np.abs(dx).max():.3e}", flush=True)
            for _ in range(3):
                prob.solve(0.01)
            p = prob.profile(reset=True)
            print(f"[{name}/{tag}] solve {p['solve_ms']/p['solve_calls']:.3f} ms", flush=True)
            t1 = time.time()
            xf, trace, rc = prob.refine(x)
            wall = time.time() - t1
            p = prob.profile(reset=True)
            print(f"[{name}/{tag}] refine rc={rc} {len(trace)} iters in {wall*1e3:.1f} ms ({wall*1e3/max(1,len(trace)):.2f} ms/iter) "
                  f"eval {p['eval_ms']:.2f} solve {p['solve_ms']:.2f} cost {p['cost_ms']:.2f}", flush=True)
            for r in trace:
                print("    ", r["iter"], f"{r['residual1']:.6e} {r['residual2']:.6e} u={r['u']:.3e} acc={r['accepted']}")
            print(f"[{name}/{tag}] gt cost {prob.cost(d['poses_gt'], True):.6e}", flush=True)
            prob.close()


if __name__ == "__main__":
    main()
