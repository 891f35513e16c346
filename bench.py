#!/usr/bin/env python
"""bench.py -- headline benchmark: LM iterations/s of the LiDAR bundle-adjustment refinement loop
(BALM2::damping_iter, reference include/BALM/bavoxel.hpp:662-767) on BASELINE.json's 2k-pose x
10M-factor synthetic problem (config C3), fp64, on N MI355X.

A "step" is one LM iteration = one trip through bavoxel.hpp:686-766: H/g/cost evaluation (if the last
step was accepted), damped LDL^T solve, retraction, cost-only evaluation, accept/reject.  When a
damping_iter run ends (<= 10 iterations or the 1e-6 relative-decrease test) the next one restarts from
the same perturbed poses, so the timed steps are the iteration mix a real refinement executes.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3|C2|NxV]
N > 1 is launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`:
one rank per GPU, voxels sharded by contiguous range (bavoxel.hpp:621-624 with thread -> GPU), the
pose-block Hessian / gradient / cost all-reduced over RCCL inside liblvba_hip.so; the two ends of the band
factorisation run on ranks 0 and 1 (the middle block is exchanged), the rest of the solve is replicated.  Total work is
fixed as N grows -> "scaling": "strong".

Rank 0 prints ONE JSON line.  `roofline` is the H/g/cost evaluation ("Jacobian") pass against HBM, from
HIP-event durations recorded on the library's stream around its kernels during the timed steps and the
algorithmic bytes of SURVEY.md section 8(d); `cpu_baseline` is the C restatement (oracle/balm_oracle.c)
timed on this host on ONE full LM iteration of the same problem, and `parity` holds the HIP path against it at that
size (the run fails when they disagree beyond 1e-7).  torch is used only for synthetic data generation on the GPU and
for the control-plane barrier; the product path is the C-ABI library.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured achievable)
FP64_PEAK_TFLOPS = 78.6    # MI355X fp64 vector == fp64 MFMA peak (256 CU x 128 flop/clk x 2.4 GHz)
L2_PEAK_TBS = 34.5         # MI355X_MICROARCH.md: aggregate L2 -> CU bandwidth, 8 XCDs
REGIONS = 5                 # timed regions of --steps steps each; the line carries the median region and the spread


def parse_config(name, synth):
    if name in synth.CONFIGS:
        return synth.CONFIGS[name]
    n, v = name.lower().split("x")
    return int(n), int(v)


def cpu_baseline_and_parity(prob, d, info):
    """The C oracle (oracle/balm_oracle.c) on the FULL problem the bench just timed -- no sampling, no scaling -- serving
    twice: (1) `cpu_baseline`: one LM iteration of the reference's loop (bavoxel.hpp:686-766) with the reference's
    threading (16-way sliced evaluation :597-639, single-threaded cost pass :176-203,731; the damped solve is the oracle's
    unpivoted band LDL^T of the REAL damped Hessian in the GPU path's pose order, where the reference runs Eigen's
    SimplicialLDLT); (2) `parity`: the HIP path against it at this size -- cost, gradient, every non-zero pose block of the
    Hessian at the initial poses, and the first LM iteration (cost before / after the step, i.e. through the solve and
    the retraction).  Returns (cpu_baseline, parity)."""
    import oracle
    N = d["n_poses"]
    co = oracle.COracle(N, d["voxel_off"], d["pose_idx"], d["clusters"])
    x0 = d["poses_init"]
    cores = min(16, os.cpu_count() or 1)               # the reference uses 16 std::threads (bavoxel.hpp:25)
    solve_threads = min(32, os.cpu_count() or 1)
    # --- HIP path
    xg, trace, rc = prob.refine(x0, max_iter=1)
    gi, gj, gblocks, g, c = prob.eval_blocks(x0)      # (sparse form: the dense Hessian is 28.8 GB at C4)
    # --- oracle: evaluation with the block list (two passes inside: count, fill)
    t = time.perf_counter()
    bi, bj, blocks, gc, cc = co.eval_sparse(x0, nthreads=cores)
    t_eval_list = 0.5 * (time.perf_counter() - t)
    h_block_rel, h_outside = oracle.block_parity_sparse(gi, gj, gblocks, bi, bj, blocks, N)
    del gblocks, blocks
    g_rel = float(np.abs(g - gc).max() / np.abs(gc).max())
    cost_rel = abs(c - cc) / abs(cc)
    # --- oracle: one LM iteration, timed per stage
    xr, tr, rcr, sec = co.damping_iter_band(x0, perm=prob.ordering(), max_iter=1, eval_threads=cores,
                                            solve_threads=solve_threads)
    lm_rel = max(abs(trace[0]["residual1"] - tr[0][1]) / tr[0][1], abs(trace[0]["residual2"] - tr[0][2]) / tr[0][2])
    pose_abs = float(np.abs(xg - xr).max())
    parity = {"against": "oracle/balm_oracle.c on the full problem", "cost_rel": cost_rel, "g_rel": g_rel,
              "H_block_rel": h_block_rel, "H_outside_pattern": h_outside, "lm_iteration_cost_rel": lm_rel,
              "poses_after_step_abs": pose_abs, "blocks_compared": int(len(bi)), "tolerance": 1e-7,
              "ok": bool(rc == 0 and rcr == 0 and max(cost_rel, g_rel, h_block_rel, lm_rel) <= 1e-7 and h_outside <= 1e-12
                         and pose_abs <= 1e-7)}
    t_iter = sec["eval"] + sec["solve"] + sec["cost"]
    n = 6 * N
    base = {
        "value": 1.0 / t_iter, "unit": "iterations/s", "cores": cores, "kind": "port",
        "sample": (f"oracle/balm_oracle.c, ONE full LM iteration of the whole problem (no sampling): H/g/cost evaluation "
                   f"{sec['eval']:.2f} s ({cores} threads, sparse block accumulation), damped solve {sec['solve']:.2f} s "
                   f"(unpivoted band LDL^T of the real damped Hessian, n = {n}, half-bandwidth {6 * sec['band_blocks'] + 5}, "
                   f"{solve_threads} threads; the reference calls Eigen::SimplicialLDLT here, single-threaded), cost-only pass "
                   f"{sec['cost']:.2f} s (1 thread, as bavoxel.hpp:176-203)"),
        "stage_s": {"eval": sec["eval"], "solve": sec["solve"], "cost": sec["cost"], "eval_with_block_list": t_eval_list},
        "host_cpus": os.cpu_count(), "host_cpu_model": cpu_model(),
    }
    return base, parity


def cpu_baseline_dense(d, base_s):
    """BASELINE.md variant (D): the reference's OWN memory scheme on the problem the bench timed -- 16 thread-local dense (6N)^2
    Hessians filled by 16 threads, summed serially in thread order, mirrored (bavoxel.hpp:603-633), then the dense D / HessuD
    matrices and the scan of all (6N)^2 entries into a triplet list (:692-703) -- timed on the host; the damped solve and the
    cost pass are the sparse port's (base_s: the reference hands the triplets to Eigen::SimplicialLDLT, which the oracle does not
    restate).  Needs ~17 x 8 (6N)^2 bytes of host memory (21 GB at C3): skipped with a note when the host has less."""
    import oracle
    N = d["n_poses"]
    n = 6 * N
    need = (17 + 3) * 8 * n * n
    avail = None
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable"):
                    avail = int(line.split()[1]) * 1024
    except Exception:
        pass
    if avail is not None and avail < 1.3 * need:
        return {"value": None, "kind": "port", "sample": f"skipped: variant (D) needs {need / 1e9:.0f} GB of host memory, {avail / 1e9:.0f} GB available"}
    co = oracle.COracle(N, d["voxel_off"], d["pose_idx"], d["clusters"])
    t0 = time.perf_counter()
    H, g, c = co.eval_dense(d["poses_init"])
    t_eval = time.perf_counter() - t0
    t0 = time.perf_counter()
    ntrip = co.dense_to_triplets(H)
    t_scan = time.perf_counter() - t0
    del H
    if ntrip < 0:
        return {"value": None, "kind": "port", "sample": "variant (D): out of host memory in the triplet scan"}
    t_iter = t_eval + t_scan + base_s["solve"] + base_s["cost"]
    return {"value": 1.0 / t_iter, "unit": "iterations/s", "cores": min(16, os.cpu_count() or 1), "kind": "port",
            "sample": (f"oracle/balm_oracle.c, variant (D) of BASELINE.md section 2 on the whole problem: 16 thread-local dense ({n} x {n}) Hessians, "
                       f"serial sum + mirror {t_eval:.2f} s (bavoxel.hpp:603-633), dense D / HessuD + scan of {n * n / 1e6:.0f} M entries into "
                       f"{ntrip / 1e6:.1f} M triplets {t_scan:.2f} s (:692-703); damped solve {base_s['solve']:.2f} s and cost pass "
                       f"{base_s['cost']:.2f} s as in `cpu_baseline` (variant S)"),
            "stage_s": {"eval_dense": t_eval, "dense_to_triplets": t_scan, "solve": base_s["solve"], "cost": base_s["cost"]},
            "host_bytes": need}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-baseline", action="store_true", help="skip the reference's own divide_thread on C2 (16 GB of host memory)")
    ap.add_argument("--no-dense-baseline", action="store_true", help="skip the CPU baseline's variant (D) at C3 (21 GB of host memory, ~20 s)")
    ap.add_argument("--no-visual", action="store_true", help="skip the (untimed-for-the-metric) visual-stage leg")
    ap.add_argument("--no-y32", action="store_true", help="skip the leg that repeats the timed steps with LVBA_Y32=1 (fp32 Y records)")
    ap.add_argument("--no-front-end", action="store_true", help="skip the (untimed-for-the-metric) voxel front-end / window-BA leg")
    ap.add_argument("--transport", choices=["rccl", "gloo"], default="rccl",
                    help="N > 1: the all-reduce of the pose blocks.  rccl = the product path (one GPU per rank).  gloo = the caller-supplied "
                         "transport entry point with a host-staged torch.distributed all-reduce: a REHEARSAL of the torchrun path on "
                         "a box with fewer GPUs than ranks (with --same-device); its rate is not a result")
    ap.add_argument("--same-device", action="store_true", help="all ranks on GPU 0 (rehearsal with --transport gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the ranks ourselves, exactly as the documented command does
        # (one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1), and become that job
        relaunch_under_torchrun(args.gpus)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_device and args.transport != "gloo" and world > 1:
        raise SystemExit("--same-device needs --transport gloo (RCCL refuses two ranks on one device)")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)   # control plane only

    pkg = importlib.import_module("global-lvba_amd")
    synth = importlib.import_module("global-lvba_amd.synth")
    N, V = parse_config(args.config, synth)
    d = synth.make_balm_problem(N, V, device=f"cuda:{local_rank}")
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    off = d["voxel_off"]
    F_total = int(off[-1])
    head, end = pkg.shard_range(V, rank, world)
    prob = pkg.BalmProblem(N, off[head:end + 1], d["pose_idx"][off[head]:off[end]], d["clusters"][off[head]:off[end]],
                           device=local_rank)
    keep_alive = None
    if world > 1 and args.transport == "gloo":
        keep_alive = gloo_allreduce_callback(dist, torch)
        prob.dist_init_external(world, rank, keep_alive[1], None)
    elif world > 1:
        uid = [pkg.BalmProblem.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        prob.dist_init(world, rank, uid[0])
    info = prob.info()
    x0 = d["poses_init"]
    prob.refine(x0)   # set-up, untimed: first-touch allocations and the solve hipGraph (captured lazily at the third solve),
                      # so that --warmup 0/1 does not put the one-off capture inside the timed steps

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    state = {"active": False, "runs": 0, "evals": 0, "accepts": 0, "evals_on_records": 0}

    def step():
        if not state["active"]:
            prob.lm_begin(x0)
            state["active"] = True
            state["runs"] += 1
        row, done, rc = prob.lm_step()
        state["evals"] += row["evaluated"]
        state["accepts"] += row["accepted"]
        # an evaluation after an accepted step starts from the voxel records the trial-point cost pass left (lvba_api.hip)
        state["evals_on_records"] += int(bool(row["evaluated"]) and row["iter"] > 0 and bool(info["trial_linearised"]))
        if done or rc != 0:
            prob.lm_end(want_poses=False)
            state["active"] = False
        return row

    for _ in range(args.warmup):
        step()
    # ---- the timed regions: REGIONS x [exactly K steps between barrier + synchronize, MAX over ranks], the library's event
    # profiling OFF (it costs ~1 % of a step).  `value` / `ms_per_step` are the MEDIAN region's; min / max are printed beside
    # them (a single 0.12-s region moved by +-1.5 % from run to run in round 4).
    prob.set_profiling(False)
    region_s, region_state = [], []
    last = None
    for _ in range(REGIONS):
        state.update(runs=int(state["active"]), evals=0, accepts=0, evals_on_records=0)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = step()
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt[0])
        region_s.append(el)
        region_state.append({k: state[k] for k in ("runs", "evals", "accepts", "evals_on_records")})
    mid = sorted(range(REGIONS), key=lambda r: region_s[r])[REGIONS // 2]
    elapsed = region_s[mid]
    state.update(region_state[mid])
    if state["active"]:
        prob.lm_end(want_poses=False)
        state["active"] = False
    timed = dict(state)
    # ---- a second, UNTIMED pass over the same K steps with HIP events on the library's stream around its kernels: the stage
    # times and the kernel durations of the `roofline` objects come from here (same problem, same start, same iteration mix)
    state.update(runs=0, evals=0, accepts=0, evals_on_records=0)
    prob.set_profiling(True)
    prob.profile(reset=True)
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed_profiled = time.perf_counter() - t1
    p = prob.profile()
    prob.set_profiling(False)
    if state["active"]:
        prob.lm_end(want_poses=False)
        state["active"] = False

    # ---- N > 1: what DOES scale over the GPUs of a node -- independent windows (the reference's window stage solves them one
    # after the other, src/lvba_system.cpp:232): every rank runs lvba_window_ba on a sequence of its own, no data exchanged;
    # reported beside the headline (whose single refinement is bound by the serial chain of its factorisation)
    windows_multi = None
    if world > 1 and not args.no_front_end:
        try:
            nw_local, dt_local = window_stage_weak(pkg, synth, rank)
        except Exception as e:
            nw_local, dt_local = 0, float("inf")
            print(f"rank {rank}: window stage failed: {e!r}", file=sys.stderr)
        tt = torch.tensor([dt_local, float(nw_local)], dtype=torch.float64)
        tmax, tsum = tt.clone(), tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        windows_multi = {"workload": "every rank: 64 scans x 250 000 points, windows of 16 (its own sequence: weak scaling)",
                         "windows": int(tsum[1]), "seconds_max_over_ranks": float(tmax[0]),
                         "windows_per_s": float(tsum[1]) / float(tmax[0]) if float(tmax[0]) > 0 else None, "scaling": "weak",
                         "note": "independent problems, no collective: lvba_window_ba per rank (one process per GPU); a single "
                                 "process drives several GPUs through lvba_window_ba_multi"}

    parity_ok = True
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        # algorithmic bytes of one H/g/cost evaluation of THIS rank's shard:
        #   84 B/factor (80 B cluster + 4 B pose index) + 4 B/voxel offsets + 96 B/pose + 8 B cost
        #   + upper pose blocks written once: 8 * (36 * nnzb + 6 N)
        Fl, Vl = info["n_factors"], info["n_voxels"]
        nnzb = prob_nnzb(prob, info)
        bytes_eval = 84 * Fl + 4 * (Vl + 1) + 96 * N + 8 + 8 * (36 * nnzb + 6 * N)
        bytes_cost = 84 * Fl + 4 * (Vl + 1) + 96 * N + 8
        ev_ms = p["eval_kernel_ms"] / max(1, p["eval_calls"])
        ck_ms = p["cost_kernel_ms"] / max(1, p["cost_calls"])
        # The trial point of a step is costed by the evaluation's own voxel pass (balm_voxel_kernel: per-chunk cost sums + the
        # voxel records); when the step is accepted the next evaluation is AT that point, starts from those records and only runs
        # the factor and pair passes.  For the roofline every evaluation is still charged with a voxel pass: the kernel time the
        # library reports for such an evaluation + one cost-stage kernel.
        eval_kernels, cost_kernels = kernel_sets(info)
        lin = bool(info["trial_linearised"])
        reuse = state["evals_on_records"] / max(1, p["eval_calls"])
        ev_ms_measured = ev_ms
        ev_ms = ev_ms + reuse * ck_ms
        sv_ms = p["solve_ms"] / max(1, p["solve_calls"])
        n = 6 * N
        bw = 6 * info["band_blocks"] + 5
        flops_solve = (n * bw * bw if info["use_band"] else n ** 3 / 3.0)
        roof = {"bound": "hbm", "kernel": "H/g/cost evaluation: " + " + ".join(eval_kernels),
                "achieved": bytes_eval / ev_ms / 1e6 if ev_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (bytes_eval / ev_ms / 1e6) / HBM_PEAK_GBS if ev_ms > 0 else None,
                "traffic": read_traffic("eval", args.config, world, eval_kernels), "traffic_source": TRAFFIC_FILE,
                "frac_of_real_traffic": None,   # filled below: the bytes the counters saw / avg_ms / peak -- how close to copy speed the passes run on what they really move
                "algorithmic_bytes": bytes_eval, "avg_ms": ev_ms,
                "avg_ms_of_evaluations_as_run": ev_ms_measured, "evaluations_on_trial_point_linearisation": reuse,
                "note": "avg_ms charges every evaluation with its first half: evaluations that start from the linearisation the "
                        "trial-point cost pass left (same kernel, same poses) are counted as their own kernels + one such pass"}
        if roof["traffic"] and ev_ms > 0:
            roof["frac_of_real_traffic"] = roof["traffic"] / ev_ms / 1e6 / HBM_PEAK_GBS
        asm_ms = max(ev_ms - ck_ms, 1e-9) if lin else None
        Ql = info["n_pairs"]
        others = [
            {"kernel": cost_kernels[0] + (" (trial-point cost pass = voxel pass of the evaluation: cost + voxel records)" if lin
                                          else " (cost-only pass)"),
             "bound": "hbm", "achieved": bytes_cost / ck_ms / 1e6,
             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_cost / ck_ms / 1e6 / HBM_PEAK_GBS,
             "traffic": read_traffic("cost", args.config, world, cost_kernels), "algorithmic_bytes": bytes_cost, "avg_ms": ck_ms},
        ]
        if others[0]["traffic"]:
            others[0]["frac_of_real_traffic"] = others[0]["traffic"] / ck_ms / 1e6 / HBM_PEAK_GBS
        if asm_ms:
            # the rest of an evaluation: factor pass (pose-major) + pair pass + partial-block sum.  Against HBM on what it must
            # read (the clusters) and write (the pose blocks, once), and against the two bounds SURVEY.md 8(d) names for the pair
            # pass: the L2 -> CU path of the two gathered 144-byte records per pair, and the fp64 vector rate of its 108 FMAs per pair.
            bytes_asm = 8 * (36 * nnzb + 6 * N) + 84 * Fl
            others.append({"kernel": "factor + pair passes: " + " + ".join(k for k in eval_kernels if k not in cost_kernels),
                           "bound": "hbm", "achieved": bytes_asm / asm_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": bytes_asm / asm_ms / 1e6 / HBM_PEAK_GBS, "traffic": None, "avg_ms": asm_ms,
                           "algorithmic_bytes": bytes_asm,
                           "note_bytes": "the clusters (84 B/factor) read once + the pose blocks written once; Y (144 B/factor "
                                         "written, then gathered per pair) is intermediate traffic, not algorithmic",
                           "l2_bound": {"bytes": 288 * Ql, "achieved": 288 * Ql / asm_ms / 1e9, "peak": L2_PEAK_TBS, "unit": "TB/s",
                                        "frac": 288 * Ql / asm_ms / 1e9 / L2_PEAK_TBS},
                           "valu_bound": {"flops": 216 * Ql, "achieved": 216 * Ql / asm_ms / 1e9, "peak": FP64_PEAK_TFLOPS,
                                          "unit": "TFLOP/s", "frac": 216 * Ql / asm_ms / 1e9 / FP64_PEAK_TFLOPS}})
        solve_roof = (
            {"kernel": "damped LDL^T solve (ldlt_step2_kernel + ldlt_diag_blocked / prepare / twist_merge / back_chain kernels)", "bound": "mfma",
             "achieved": flops_solve / sv_ms / 1e9, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
             "frac": flops_solve / sv_ms / 1e9 / FP64_PEAK_TFLOPS, "traffic": None,
             "algorithmic_flops": flops_solve, "avg_ms": sv_ms,
             # the second bound: every C tile of a panel's window is read and written once per PAIR of 64-column panels
             # (rank-128 update: 2 * 128 flops per 16 bytes moved), so HBM caps the trailing updates at 16 flop/B
             "hbm_bound": {"flop_per_byte": 16.0, "peak": 16.0 * HBM_PEAK_GBS / 1e3, "unit": "TFLOP/s",
                           "frac": flops_solve / sv_ms / 1e9 / (16.0 * HBM_PEAK_GBS / 1e3),
                           "note": "above the MFMA peak since the panels are paired (8 flop/B -> 64 TFLOP/s with "
                                   "single panels): the matrix pipe and the serial chain of panel factorisations "
                                   "bound the solve, not HBM"}})
        others.append(solve_roof)
        out = {
            "metric": "LM iterations/sec, 2k poses x 10M LiDAR factors (BALM damping_iter)",
            "value": args.steps / elapsed, "unit": "iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: BALM LiDAR BA, {N} poses x {V} voxels x {F_total} LiDAR factors "
                                   f"(plane-eigenvalue factors, exact Hessian, Nielsen LM)",
                       "n_poses": N, "n_voxels": V, "n_factors": F_total, "n_pairs_local": info["n_pairs"],
                       "sharding": (f"voxel ranges over {world} rank(s), {'RCCL' if args.transport == 'rccl' else 'HOST-STAGED gloo (rehearsal)'} all-reduce of pose-block H/g/cost, "
                                    f"{info['allreduce_bytes'] / 1e6:.0f} MB per evaluation") if world > 1 else "single GPU",
                       "solver": ("band" if info["use_band"] else "dense") + f" LDL^T, half-bandwidth {bw} of n={n}",
                       "lm_runs": timed["runs"], "evals_in_timed_steps": timed["evals"],
                       "accepted_in_timed_steps": timed["accepts"], "last_cost": last["residual2"] if last else None},
            "stage_ms": {"eval": p["eval_ms"] / max(1, p["eval_calls"]), "solve": sv_ms,
                         "cost": p["cost_ms"] / max(1, p["cost_calls"]),
                         "allreduce": p["reduce_ms"] / max(1, p["reduce_calls"]) if p["reduce_calls"] else 0.0},
            "stage_ms_source": {"pass": "second, untimed pass over the same K steps with the library's HIP-event profiling on "
                                        "(the timed region runs with it off)", "ms_per_step": 1e3 * elapsed_profiled / args.steps,
                                "evals": state["evals"], "accepted": state["accepts"]},
            "timing": {"regions": REGIONS, "steps_per_region": args.steps, "value_is": "median region",
                       "ms_per_step_by_region": [1e3 * t / args.steps for t in region_s],
                       "ms_per_step_min": 1e3 * min(region_s) / args.steps, "ms_per_step_max": 1e3 * max(region_s) / args.steps,
                       "value_min": args.steps / max(region_s), "value_max": args.steps / min(region_s),
                       "median_region": mid,
                       # (the regions continue one another's LM runs: their iteration mix differs, deterministically)
                       "evals_by_region": [r["evals"] for r in region_state],
                       "accepted_by_region": [r["accepts"] for r in region_state],
                       "lm_runs_started_by_region": [r["runs"] for r in region_state]},
            "roofline": roof,
            # the kernel that dominates the step BY TIME (the evaluation above is the one north_star puts a number on)
            "roofline_solve": solve_roof,
            # the LM loop's own cost pass (every iteration runs one at the trial point), as timed inside the loop
            "roofline_cost_pass_in_loop": others[0],
            "roofline_other_kernels": others,
            "scaling_model": scaling_model(p, sv_ms, info, world, prob),
        }
        if windows_multi is not None:
            out["window_stage_all_ranks"] = windows_multi
        if world == 1 and not args.no_y32:
            try:
                out["y32_mode"] = y32_leg(pkg, d, N, prob, local_rank, args.steps, args.warmup)
            except Exception as e:
                out["y32_mode"] = {"error": repr(e)}
        if world == 1 and not args.no_visual:
            try:
                out["visual_stage"] = visual_leg(pkg, synth, N, local_rank, not args.no_cpu_baseline)
            except Exception as e:
                out["visual_stage"] = {"error": repr(e)}
        if world == 1 and not args.no_front_end:
            try:
                out["front_end"] = front_end_leg(pkg, synth, not args.no_cpu_baseline)
            except Exception as e:
                out["front_end"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline and not args.no_reference_baseline:
            try:
                out["cpu_baseline_reference_c2"] = lidar_reference_baseline_c2(pkg, synth, local_rank)
            except Exception as e:
                out["cpu_baseline_reference_c2"] = {"value": None, "kind": "reference", "sample": f"failed: {e!r}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity(prob, d, info)
                if args.config == "C3" and not args.no_dense_baseline:
                    try:
                        out["cpu_baseline_dense_c3"] = cpu_baseline_dense(d, out["cpu_baseline"]["stage_s"])
                    except Exception as e:
                        out["cpu_baseline_dense_c3"] = {"value": None, "kind": "port", "sample": f"failed: {e!r}"}
                parity_ok = out["parity"]["ok"]
            except Exception as e:  # the baseline is a reported number, not part of the measured path
                out["cpu_baseline"] = {"value": None, "unit": "iterations/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)
    prob.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not parity_ok:
        raise SystemExit("bench.py: the HIP path disagrees with the oracle beyond 1e-7 at the benchmark size (see \"parity\")")


def relaunch_under_torchrun(n):
    """Replace this process by `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same arguments>`
    (never returns).  The port is one the kernel has just handed out as free."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs across processes on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    sys.stdout.flush(); sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


def scaling_model(p, sv_ms, info, world, prob=None):
    """What strong scaling of ONE refinement can give, from this run's own stage times: the evaluation and the cost pass shard by
    voxel range (/ N), the pose blocks are all-reduced (2 (N - 1) / N x bytes / an ASSUMED 250 GB/s RCCL bus bandwidth over xGMI -- 7
    links x ~153 GB/s per GPU, full mesh; never measured here: no multi-GPU node was available to any round), and the
    damped solve is a serial chain of panel factorisations that two ranks split at best (the two ends of the band) -- with both
    ends already sharing every launch on one GPU, the second rank buys almost nothing.  A projection, not a measurement."""
    # (a run on `world` ranks measured its shards: the one-GPU stage times are ~world x those)
    ev = world * p["eval_ms"] / max(1, p["eval_calls"])
    ck = world * p["cost_ms"] / max(1, p["cost_calls"])
    ar_mb = 8e-6 * (36 * (info.get("n_blocks", 0) + info["n_poses"]) + 6 * info["n_poses"] + 1)
    bus, bus_src = 250.0, "assumed (no multi-GPU node was available to any round); tools/first_node.sh measures it"
    try:   # the measured figure, once a node has been seen (tools/rccl_smoke.py writes it)
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "rccl_bus_bandwidth.json")) as f:
            mb = json.load(f)
        bus, bus_src = float(mb["bus_gb_s"]), f"measured by tools/rccl_smoke.py on {mb['n_gpus']} GPUs (profiles/rccl_bus_bandwidth.json)"
    except Exception:
        pass
    proj = {}
    for n in (1, 2, 4, 8):
        ar = 0.0 if n == 1 else 2.0 * (n - 1) / n * ar_mb / bus   # ms at `bus` GB/s bus bandwidth (MB / (GB/s) = ms)
        sv, how = sv_ms, "band LDL^T as measured on this GPU (a second rank takes the other end: no gain over both ends in one launch)"
        nd = None
        if n > 1 and prob is not None:
            # the solver's own plan for n ranks (lvba_balm_nd_model: csrc/nd_plan.h on this problem's co-visibility graph): arcs of the
            # band dealt out over the ranks + a separator system, costed in the solver's measured units -- a MODEL (its band figure
            # next to the measured solve says how far off it is); taken when it beats the band
            try:
                nd = prob.nd_model(n)
            except Exception:
                nd = None
            if nd and nd["nd_ms"] > 0 and nd["nd_ms"] * (sv_ms / max(nd["band_ms"], 1e-9)) < sv_ms:
                sv = nd["nd_ms"] * (sv_ms / nd["band_ms"])   # scaled by measured / modelled band time of this very problem
                how = (f"nested dissection over {n} ranks (csrc/ldlt_nd.h): {nd['arcs']} arcs of <= {nd['max_arc_poses']} poses, separator "
                       f"{nd['sep_poses']} poses; model {nd['nd_ms']:.2f} ms against {nd['band_ms']:.2f} ms for the band, scaled by this "
                       f"run's measured / modelled band solve")
        t = (ev + ck) / n + ar + sv
        proj[str(n)] = {"ms_per_iteration": t, "speedup": (ev + ck + sv_ms) / t, "solve_ms": sv, "solve": how, "nd_model": nd}
    return {"kind": "strong scaling of one refinement: evaluation / cost pass sharded by voxel range, pose blocks all-reduced, solve by the band (two ranks at best) or by nested dissection over the ranks -- whichever the solver's model prefers", "stage_ms_1gpu": {"eval": ev, "cost": ck, "solve": sv_ms}, "stage_ms_1gpu_from": f"this run's stage times on {world} rank(s), evaluation and cost pass scaled by the rank count",
            "allreduce_mb_per_evaluation": ar_mb, "bus_gb_s": bus, "bus_gb_s_source": bus_src, "projected": proj,
            "note": "a projection, not a measurement (no multi-GPU node was available to any round).  A band that is short compared "
                    "with its width (C3: n / bw = 4.6) is a serial chain of panel factorisations that does not shard -- throughput "
                    "across GPUs then comes from independent work: windows (window_stage_all_ranks, lvba_window_ba_multi), "
                    "sequences; a long band (C4: n / bw = 24) is dissected into one arc per rank (csrc/nd_plan.h)",
            "measured_ranks": world}


def window_stage_weak(pkg, synth, rank):
    """one rank's share of the weak-scaling window leg: (windows, seconds) of lvba_window_ba on its own 64-scan sequence"""
    base = synth.make_scans(8, 250_000, room=(60, 40, 8), n_panels=16, n_blobs=40, origin=(120.0 + 7.0 * rank, -80.0, 2.0),
                            rot_sigma_deg=0.1, trans_sigma=0.03, point_floats=12, seed=100 + rank)
    clouds, poses = [], []
    for r in range(8):
        for c, T in zip(base["clouds"], base["poses"]):
            T = T.copy(); T[9] += 100.0 * r
            clouds.append(c); poses.append(T)
    poses = np.asarray(poses)
    import torch
    scans = pkg.Scans(clouds, device=torch.cuda.current_device())
    scans.window_ba(poses, window_size=16, voxel_size=0.5, anchor_leaf=0.05)["anchor_scans"].close()    # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w = scans.window_ba(poses, window_size=16, voxel_size=0.5, anchor_leaf=0.05)
    dt = time.perf_counter() - t0
    w["anchor_scans"].close()
    scans.close()
    return len(w["windows"]), dt


def y32_leg(pkg, d, N, prob_ref, local_rank, steps, warmup):
    """The same K timed steps with LVBA_Y32=1: the per-factor Y records travel as fp32 (80 instead of 144 bytes) between the
    factor pass and the pair pass -- off-diagonal Hessian blocks then carry ~6e-7 relative rounding, the cost and the gradient
    none.  Reported BESIDE the headline (which runs the fp64 records), with what north_star judges: every LM cost of a whole
    refinement and the refined poses against the fp64-record run of the same handle type (bar 1e-5)."""
    x0 = d["poses_init"]
    x_ref, tr_ref, _ = prob_ref.refine(x0)
    old = os.environ.get("LVBA_Y32")
    os.environ["LVBA_Y32"] = "1"
    try:
        prob = pkg.BalmProblem(N, d["voxel_off"], d["pose_idx"], d["clusters"], device=local_rank)
        info = prob.info()       # (the switch is read when the handle is finalised: at its first use, i.e. here)
    finally:
        if old is None:
            os.environ.pop("LVBA_Y32", None)
        else:
            os.environ["LVBA_Y32"] = old
    x, tr, rc = prob.refine(x0)
    prob.refine(x0)   # (solve graph captured)
    active = [False]

    def step():
        if not active[0]:
            prob.lm_begin(x0)
            active[0] = True
        row, done, rc2 = prob.lm_step()
        if done or rc2 != 0:
            prob.lm_end(want_poses=False)
            active[0] = False

    import torch
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if active[0]:
        prob.lm_end(want_poses=False)
    prob.set_profiling(True)
    prob.profile(reset=True)
    prob.refine(x0)
    p = prob.profile()
    prob.close()
    n = min(len(tr), len(tr_ref))
    cost_rel = max(max(abs(tr[k]["residual1"] - tr_ref[k]["residual1"]) / abs(tr_ref[k]["residual1"]),
                       abs(tr[k]["residual2"] - tr_ref[k]["residual2"]) / abs(tr_ref[k]["residual2"])) for k in range(n))
    return {"switch": "LVBA_Y32=1", "took_effect": bool(info["y_fp32"]), "ms_per_step": 1e3 * dt / steps, "value": steps / dt,
            "unit": "iterations/s", "eval_ms": p["eval_ms"] / max(1, p["eval_calls"]),
            "parity_vs_fp64_records": {"lm_iterations": [len(tr_ref), len(tr)], "lm_cost_rel_max": cost_rel,
                                       "accept_pattern_equal": [r["accepted"] for r in tr[:n]] == [r["accepted"] for r in tr_ref[:n]],
                                       "final_pose_abs_max": float(np.abs(x - x_ref).max()), "tolerance": 1e-5,
                                       "ok": bool(rc == 0 and len(tr) == len(tr_ref) and cost_rel <= 1e-5 and np.abs(x - x_ref).max() <= 1e-5)},
            "note": "not the headline: `value` above runs fp64 Y records; this mode stores an intermediate in fp32 (fp64 accumulation)"}


def gloo_allreduce_callback(dist, torch):
    """An lvba_allreduce_fn (include/lvba_hip.h) in Python: device buffer -> host, torch.distributed (gloo) all-reduce, host ->
    device.  For rehearsing the torchrun path of this script with more ranks than GPUs; returns (callback object, its address)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    kinds = {0: (8, torch.float64), 1: (8, torch.int64), 2: (4, torch.int32), 3: (1, torch.uint8)}
    FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_void_p)

    def cb(ctx, buf, count, dtype, op, stream):
        try:
            nb, td = kinds[dtype]
            host = torch.empty(int(count), dtype=td)
            if hip.hipStreamSynchronize(stream) or hip.hipMemcpy(host.data_ptr(), buf, int(count) * nb, 2):   # device -> host
                return 1
            # every rank sums the ranks' buffers in rank order: bitwise the same result everywhere (what the header asks for)
            parts = [torch.empty_like(host) for _ in range(dist.get_world_size())]
            dist.all_gather(parts, host)
            acc = parts[0]
            for q in parts[1:]:
                acc = torch.maximum(acc, q) if op == 1 else acc + q
            return 1 if hip.hipMemcpy(buf, acc.data_ptr(), int(count) * nb, 1) else 0                         # host -> device
        except Exception:
            return 1
    fn = FN(cb)
    return fn, C.cast(fn, C.c_void_p).value


def rms(v):
    return float(np.sqrt((np.asarray(v) ** 2).sum(-1).mean()))


def quat_to_rot(q):
    w, x, y, z = (np.asarray(q) / np.linalg.norm(q, axis=1, keepdims=True)).T
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
                     np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
                     np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], -2)


def cam_centres(q, t):
    """camera centres -R_cw^T t_cw from q [M,4] (w,x,y,z) and t [M,3]"""
    return -np.einsum("nji,nj->ni", quat_to_rot(q), np.asarray(t))


def rot_err_deg(q, q_gt):
    R = np.einsum("nij,nkj->nik", quat_to_rot(q), quat_to_rot(q_gt))
    c = np.clip((np.trace(R, axis1=1, axis2=2) - 1.0) / 2.0, -1.0, 1.0)
    return float(np.degrees(np.sqrt((np.arccos(c) ** 2).mean())))


def visual_leg(pkg, synth, n_cams, local_rank, with_cpu=True):
    """The second, separate problem of config C3: the visual stage (500k reprojection observations on 125k landmarks,
    cameras = the 2k poses), solved after the LiDAR stage as the reference does (src/lvba_system.cpp:139-140).  Reported
    beside the headline metric, never inside `value`.  `roofline`: the factor kernels (residual / Jacobian / Schur products)
    against HBM on their algorithmic bytes (24 B/observation + 56 B/landmark + 56 B/camera + the reduced
    system written once), and the reduced-camera-system solve against its own bound (a 143-wide band: the serial panel chain,
    not flops).  `cpu_baseline`: the reference's own cost functors, differentiated with Jets as ceres::AutoDiffCostFunction
    does, over all observations on all host cores (oracle/_ref, kind "reference"), plus a multi-threaded LAPACK Cholesky of a
    dense SPD matrix of the reduced system's size -- what Ceres' DENSE_SCHUR factorises every iteration; the Schur elimination
    itself is left out, so the figure is an UPPER bound of the CPU's iteration rate (restatement, not Ceres)."""
    # initial values OUTSIDE what the data can resolve (0.3 deg, 10 cm per camera, 30 cm per landmark; the defaults of the
    # generator -- 0.05 deg, 2 cm, 5 cm -- are below the depth resolution of 0.5 px over a 1.8 m baseline): the refinement
    # must bring every error against ground truth DOWN
    d = synth.make_visual_problem(n_cams, 125_000, rot_sigma_deg=0.3, trans_sigma=0.10, point_sigma=0.30, device=f"cuda:{local_rank}")
    prob = pkg.VisualProblem(n_cams, d["obs_off"], d["obs_cam"], d["obs_uv"], d["plane"], d["valid"], d["intr"], device=local_rank)
    prob.refine(d["q"], d["t"], d["X"], max_iter=3)                      # warm-up (graph capture, allocations)
    t0 = time.perf_counter()
    (q, t, X), trace, term, rc = prob.refine(d["q"], d["t"], d["X"])
    dt = time.perf_counter() - t0
    iters = max(1, len(trace) - 1)
    # the cost of ONE iteration inside the loop (no state upload, first evaluation, Jacobi scaling, download): the slope between a
    # refinement capped at two iterations and the whole one
    t0 = time.perf_counter()
    _, trace2, _, _ = prob.refine(d["q"], d["t"], d["X"], max_iter=2)
    dt2 = time.perf_counter() - t0
    it2 = max(1, len(trace2) - 1)
    in_loop_ms = 1e3 * (dt - dt2) / (iters - it2) if iters > it2 else None
    # the factor kernels alone: residuals + Jacobians + column norms + Schur products -> reduced system (no solve, no download)
    lin = []
    for _ in range(5):
        t1 = time.perf_counter()
        prob.linearize_only(d["q"], d["t"], d["X"])
        lin.append(time.perf_counter() - t1)
    lin_ms = 1e3 * min(lin)
    n_obs = int(d["obs_off"][-1])
    vmask = np.asarray(d["valid"]) != 0                                   # landmarks without a plane are left out entirely
    n_obs_act = int(np.diff(d["obs_off"])[vmask].sum())
    n_res = 2 * n_obs_act + int(vmask.sum())                               # reprojection (2 / observation) + plane priors
    n = 6 * n_cams
    info = prob.info()
    bw = 6 * info["band_blocks"] + 5
    bytes_factor = 24 * n_obs_act + 56 * int(vmask.sum()) + 56 * n_cams + 8 * (36 * (info["n_blocks"] + n_cams) + n)
    iter_ms = 1e3 * dt / iters
    solve_ms = max(iter_ms - lin_ms, 1e-6)     # the rest of an iteration: band solve, back-substitution, step, cost at the trial point
    flops_solve = n * bw * bw
    out = {"workload": f"{n_cams} cameras x 125000 landmarks x {n_obs} reprojection observations + plane priors",
           "lm_iterations": iters, "iterations_per_s": iters / dt, "ms_per_iteration": iter_ms, "termination": term,
           "ms_per_iteration_in_loop": in_loop_ms,
           "fixed_ms": (1e3 * dt - iters * in_loop_ms) if in_loop_ms is not None else None,
           "observations_per_s": n_obs_act * iters / dt,
           "stage_ms": {"linearize (stand-alone call: state upload + factor kernels)": lin_ms, "rest of an iteration": solve_ms},
           "stage_note": "a stand-alone linearisation re-uploads the state (~0.23 ms of its time); inside the LM loop the state stays on the "
                         "device -- see ms_per_iteration_in_loop, and LVBA_TIMING=vis (tools/visual_bench.py) for the phases of an iteration",
           "roofline": {"bound": "hbm", "kernel": "vis_residual / vis_colnorm / vis_point / vis_cam / pair pass (one linearisation)",
                        "achieved": bytes_factor / lin_ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": bytes_factor / lin_ms / 1e6 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": bytes_factor,
                        "avg_ms": lin_ms,
                        "note": "19 MB per linearisation: launch- and latency-bound, not bandwidth-bound (SURVEY.md 8(d))"},
           "roofline_solve": {"bound": "mfma", "kernel": "reduced camera system: band LDL^T from both ends (ldlt_* kernels)",
                              "achieved": flops_solve / solve_ms / 1e9, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": flops_solve / solve_ms / 1e9 / FP64_PEAK_TFLOPS, "algorithmic_flops": flops_solve,
                              "avg_ms": solve_ms, "half_bandwidth": bw, "n": n,
                              "note": f"a {bw}-wide band has {flops_solve / 1e9:.2f} GFLOP: the bound is the serial chain of "
                                      f"{ldlt_chain(n, bw)} 64-column panels (~25 us each), not the matrix pipe"},
           "cost_initial": trace[0]["cost"], "cost_final": trace[-1]["cost"],
           # against the planted solution: camera CENTRES -R^T t (t_cw alone carries the rotation error times the distance from
           # the world origin, ~100 m here) and rotation angles; landmarks that take part (those with a plane)
           "camera_centre_err_m": {"initial_rms": rms(cam_centres(d["q"], d["t"]) - cam_centres(d["q_gt"], d["t_gt"])),
                                   "final_rms": rms(cam_centres(q, t) - cam_centres(d["q_gt"], d["t_gt"]))},
           "camera_rotation_err_deg": {"initial_rms": rot_err_deg(d["q"], d["q_gt"]), "final_rms": rot_err_deg(q, d["q_gt"])},
           "landmark_err_m": {"initial_rms": rms((d["X"] - d["X_gt"])[vmask]), "final_rms": rms((X - d["X_gt"])[vmask])},
           "residual_rms_whitened": {"initial": float(np.sqrt(2 * trace[0]["cost"] / n_res)),
                                     "final": float(np.sqrt(2 * trace[-1]["cost"] / n_res))},
           "note": "planted-solution recovery: initial errors 0.3 deg / 10 cm / 30 cm; what remains after the refinement is what "
                   "0.5 px observations over a 4-camera, 1.8 m baseline (depth sigma ~ z^2 sigma_px / (f b)) and 1 cm plane "
                   "priors resolve"}
    if with_cpu:
        try:
            out["cpu_baseline"] = visual_cpu_baseline(d, n, trace[0]["cost"])
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "kind": "reference", "sample": f"failed: {e!r}"}
    return out


def ldlt_chain(n, bw):
    """serial panels of the two-ended band factorisation (csrc/ldlt.hip: P + |S| / 64)"""
    P = max(0, (n - bw) // 128)
    if P < 4:
        return -(-n // 64)
    return P + -(-(n - 128 * P) // 64)


def visual_cpu_baseline(d, n, gpu_cost0):
    import oracle
    cores = os.cpu_count() or 1
    if not oracle.Reference.available():
        raise RuntimeError("oracle/_ref/libbalm_ref.so is absent")
    ref = oracle.Reference()
    ref.visual_jacobian_pass(d["q"], d["t"], d["X"], d["obs_off"], d["obs_cam"], d["obs_uv"], d["plane"], d["valid"], d["intr"],
                             nthreads=cores)
    t_jac, cost = ref.visual_jacobian_pass(d["q"], d["t"], d["X"], d["obs_off"], d["obs_cam"], d["obs_uv"], d["plane"], d["valid"],
                                           d["intr"], nthreads=cores)
    # dense Cholesky of an SPD matrix of the reduced camera system's size (what DENSE_SCHUR factorises each iteration)
    rng = np.random.default_rng(0)
    A = rng.standard_normal((n, 64))
    S = A @ A.T
    S[np.diag_indices(n)] += n
    t0 = time.perf_counter()
    np.linalg.cholesky(S)
    t_chol = time.perf_counter() - t0
    return {"value": 1.0 / (t_jac + t_chol), "unit": "iterations/s", "cores": cores, "kind": "reference",
            "sample": (f"one LM iteration's two largest stages on the whole problem: residual + Jet-Jacobian evaluation with the "
                       f"reference's own functors (include/utils.hpp, oracle/_ref/libbalm_ref.so, OpenMP x{cores}) {t_jac:.3f} s; "
                       f"dense Cholesky of a {n} x {n} SPD matrix (numpy / LAPACK, multi-threaded; Ceres' DENSE_SCHUR does this with "
                       f"Eigen, single-threaded) {t_chol:.3f} s; the Schur elimination between them is not timed, so this is an upper "
                       f"bound of the CPU rate (restatement, not Ceres)"),
            "stage_s": {"jacobian": t_jac, "dense_cholesky": t_chol},
            "parity_cost_rel": abs(cost - gpu_cost0) / abs(cost), "host_cpus": cores, "host_cpu_model": cpu_model()}


def lidar_reference_baseline_c2(pkg, synth, local_rank):
    """kind "reference": the reference's OWN BALM2::divide_thread and only_residual (include/BALM/bavoxel.hpp compiled against
    the stand-ins of oracle/shim, oracle/_ref/libbalm_ref.so) on config C2 in the reference's literal layout -- a dense
    win_size-slot PointCluster array per voxel (400k x 500 x 80 B = 16 GB) and 16 thread-local dense (6N)^2 Hessians.  C3 cannot
    be represented in that layout (416 GB of slots), which is why the headline's cpu_baseline is the sparse port.  The damped
    solve is not timed here: the reference calls Eigen::SimplicialLDLT, of which oracle/shim holds only a stand-in."""
    import oracle
    if not oracle.Reference.available():
        return {"value": None, "kind": "reference", "sample": "oracle/_ref/libbalm_ref.so is absent"}
    try:
        with open("/proc/meminfo") as f:
            avail_gb = [int(l.split()[1]) for l in f if l.startswith("MemAvailable")][0] / 1e6
    except Exception:
        avail_gb = 0.0
    N, V = synth.CONFIGS["C2"]
    if avail_gb < 40.0:
        return {"value": None, "kind": "reference", "sample": f"skipped: {avail_gb:.0f} GB of host memory available, 40 GB needed"}
    import torch
    d = synth.make_balm_problem(N, V, device=f"cuda:{local_rank}")
    torch.cuda.empty_cache()
    off, idx = d["voxel_off"], d["pose_idx"]
    slots = np.zeros((V, N, 10))
    slots[np.repeat(np.arange(V), np.diff(off)), idx] = d["clusters"]
    ref = oracle.Reference()
    x = d["poses_init"]
    t0 = time.perf_counter(); H, g, r = ref.divide_thread(slots, x); t_eval = time.perf_counter() - t0
    t0 = time.perf_counter(); c = ref.only_residual(slots, x, is_avg=True); t_cost = time.perf_counter() - t0
    del slots
    prob = pkg.BalmProblem(N, off, idx, d["clusters"], device=local_rank)
    Hg, gg, cg = prob.eval(x)
    prob.set_profiling(True); prob.profile(reset=True)
    for _ in range(5):
        prob.eval(x, want_H=False, want_g=False); prob.cost(x)
    p = prob.profile()
    prob.close()
    return {"workload": f"C2: {N} poses x {V} voxels x {int(off[-1])} factors", "kind": "reference", "cores": 16,
            "unit": "s per call",
            "divide_thread_s": t_eval, "only_residual_s": t_cost,
            "hip_eval_ms": p["eval_ms"] / max(1, p["eval_calls"]), "hip_cost_ms": p["cost_ms"] / max(1, p["cost_calls"]),
            "parity_vs_reference": {"cost_rel": abs(cg - r) / abs(r), "g_rel": float(np.abs(gg - g).max() / np.abs(g).max()),
                                    "H_rel": float(np.abs(Hg - H).max() / np.abs(H).max())},
            "sample": "oracle/_ref/libbalm_ref.so: BALM2::divide_thread (16 std::threads, dense slots, dense per-thread Hessians, "
                      "bavoxel.hpp:597-639) and BALM2::only_residual (1 thread, :641-648) on the whole C2 problem",
            "host_cpus": os.cpu_count(), "host_cpu_model": cpu_model()}


def front_end_leg(pkg, synth, with_cpu):
    """The step BEFORE the refinement: raw fp32 scans -> adaptive-voxel plane map -> packed problem
    (lvba_voxmap_build_scans), and the window-BA stage (lvba_window_ba).  Scans resident in HBM when the clock starts.
    Reported beside the headline metric, never inside `value`; the CPU figure is the C++ restatement with the reference's
    data structures (oracle/voxel_oracle.cpp) on the ray-cast base frames, one core."""
    base = synth.make_scans(8, 250_000, room=(60, 40, 8), n_panels=16, n_blobs=40, origin=(120.0, -80.0, 2.0),
                            rot_sigma_deg=0.1, trans_sigma=0.03, point_floats=12)
    clouds, poses = [], []
    for r in range(8):                      # 64 scans: the ray-cast room repeated every 100 m along x
        for c, T in zip(base["clouds"], base["poses"]):
            T = T.copy(); T[9] += 100.0 * r
            clouds.append(c); poses.append(T)
    poses = np.asarray(poses)
    out = {"workload": f"{len(clouds)} scans x {len(clouds[0])} points (48-byte PCL stride on the host), 1.0 m root voxels"}
    ups = []
    scans = None
    for _ in range(2):     # (the first upload of a process also pays for the runtime's first pinned allocations: 17-20 ms became 85 on one box)
        if scans is not None:
            scans.close()
        t0 = time.perf_counter()
        scans = pkg.Scans(clouds)
        ups.append(1e3 * (time.perf_counter() - t0))
    out["upload_ms"] = min(ups)
    out["upload_ms_first_call"] = ups[0]
    best, m = 1e9, None
    for _ in range(4):
        if m is not None:
            m.close()
        t0 = time.perf_counter()
        m = scans.voxel_map(poses, 1.0)
        best = min(best, time.perf_counter() - t0)
    npts = m.info["n_points"]
    out.update({"map_ms": 1e3 * best, "points_per_s": npts / best,
                "points_per_s_end_to_end": npts / (best + 1e-3 * out["upload_ms"]),   # host clouds (48-byte stride) -> plane map
                "n_points": npts, "n_plane_voxels": m.info["n_voxels"],
                "n_factors": m.info["n_factors"],
                "phase_ms": {k: m.info[k] for k in ("key_ms", "sort_ms", "count_ms", "write_ms")}})
    # The bound of the map build is HBM (integer / byte work, no contraction).  ALGORITHMIC bytes: every point read once (12 B),
    # the poses, and what lvba_balm_create takes written once (80 B cluster + 4 B pose index per factor, 8 B offset per voxel,
    # 48 B per plane).  `passes_bytes_model` is what the sort-based formulation moves by construction (csrc/voxelize.hip: key +
    # record + index written, 32-bit key re-pack, onesweep histogram + 8-bit digit passes over (key, index), record gather, head
    # flags, two scans, table pass, cluster pass) -- the ratio to the algorithmic bytes is the price of the sort.
    nf, nv, npl = m.info["n_factors"], m.info["n_voxels"], m.info.get("n_planes", m.info["n_voxels"])
    fe_bytes = 12 * npts + 96 * len(clouds) + 84 * nf + 8 * (nv + 1) + 48 * npl
    digit_passes = 3
    per_point_model = 40 + 12 + (4 + 16 * digit_passes) + 48 + 32 + 16 + 40 + 16
    out["roofline"] = {"bound": "hbm", "kernel": "lvba_voxmap_build_scans: vox_key + rocprim onesweep + vox_gather + vox_heads / scans / "
                                                 "tables + vox_seg_* / vox_root / vox_split_* + vox_*_emit",
                       "achieved": fe_bytes / best / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": fe_bytes / best / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": fe_bytes, "avg_ms": 1e3 * best,
                       "traffic": None, "passes_bytes_model": per_point_model * npts,
                       "frac_of_model_bytes": per_point_model * npts / best / 1e9 / HBM_PEAK_GBS,
                       "note": "wall time of the whole call (host-side allocation, eight stream synchronisations to read counts "
                               "back, launches) -- not the sum of kernel durations"}
    m.close()
    scans.window_ba(poses, window_size=16, voxel_size=0.5, anchor_leaf=0.05)["anchor_scans"].close()    # warm-up
    t0 = time.perf_counter()
    w = scans.window_ba(poses, window_size=16, voxel_size=0.5, anchor_leaf=0.05)
    dt = time.perf_counter() - t0
    nw = len(w["windows"])
    pts_w = npts / nw
    n_anchor = float(np.mean([x["n_anchor_points"] for x in w["windows"]]))
    # per window: its scans are read twice (voxel map; alignment + anchor merge), the down-sampled anchor cloud is written once
    wb_bytes = 2 * 12 * pts_w + 12 * n_anchor
    out["window_ba"] = {"windows": nw, "frames_per_window": 16, "ms_per_window": 1e3 * dt / nw,
                        "skipped": int(sum(x["skipped"] for x in w["windows"])),
                        "lm_iterations": [x["n_iter"] for x in w["windows"]],
                        # the library's own host clocks around the stages of a window (lvba_window_info_t; joint stages are
                        # shared out evenly over the windows they served): voxel map, LM set-up (create, pair lists, ordering),
                        # LM (set-up included), alignment + anchor merge
                        "stage_ms_per_window": {k: float(np.mean([x[k] for x in w["windows"]]))
                                                for k in ("map_ms", "setup_ms", "solve_ms", "merge_ms")},
                        "roofline": {"bound": "hbm", "kernel": "lvba_window_ba (voxel map + grouped LM + anchor merge), per window",
                                     "achieved": wb_bytes * nw / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": wb_bytes * nw / dt / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": wb_bytes,
                                     "avg_ms": 1e3 * dt / nw, "traffic": None,
                                     "note": "latency-bound by construction: a window is 16 x 250 k points and a 96-unknown LM "
                                             "problem; the bound that matters is the launch / host turn-around count"}}
    w["anchor_scans"].close()
    scans.close()
    if with_cpu:
        import oracle
        ref = oracle.voxel_build_cpp([c[:, :3] for c in base["clouds"]], base["poses"], 1.0)
        n = sum(len(c) for c in base["clouds"])
        out["cpu_baseline"] = {"value": n / ref["seconds"], "unit": "points/s", "cores": 1, "kind": "port",
                               "sample": f"oracle/voxel_oracle.cpp on the {len(base['clouds'])} ray-cast base scans ({n} points), "
                                         f"{ref['seconds']:.2f} s"}
        try:   # the reference's own cut_voxel / recut / tras_opt (oracle/_ref: bavoxel.hpp compiled against the stand-ins of
               # oracle/shim; the hash-map / octree work that dominates here is the reference's own code), when it is available
            if oracle.Reference.available():
                m = oracle.Reference().map_build([c[:, :3] for c in base["clouds"]], base["poses"], 1.0)
                oracle.Reference().map_free(m["handle"])
                out["cpu_baseline_reference"] = {
                    "value": n / m["seconds"], "unit": "points/s", "cores": 1, "kind": "reference",
                    "sample": f"oracle/_ref/libbalm_ref.so (the reference's include/BALM/bavoxel.hpp: cut_voxel + recut + tras_opt) on "
                              f"the same {len(base['clouds'])} scans ({n} points), {m['seconds']:.2f} s"}
        except Exception as e:  # the baseline is reporting only: never let it take the bench down
            out["cpu_baseline_reference"] = {"value": None, "kind": "reference", "sample": f"failed: {e!r}"}
    return out


def prob_nnzb(prob, info):
    """distinct lower pose blocks incl. diagonal (== upper) that the evaluation writes"""
    return info.get("n_blocks", 0) + info["n_poses"]


# the kernels the `roofline` entries cover; a committed PMC summary is only quoted when it was taken from these very kernels
def kernel_sets(info):
    """(kernels of one H/g/cost evaluation, kernels of the LM loop's trial-point cost pass) as the library ran them
    (lvba_balm_info: trial_linearised)."""
    ev = ["balm_voxel_kernel", "balm_factor_kernel", "balm_diag_reduce_kernel", "balm_pair_col_kernel", "balm_pair_reduce_kernel"]
    return ev, (["balm_voxel_kernel"] if info["trial_linearised"] else ["balm_cost_kernel"])


TRAFFIC_FILE = os.path.join("profiles", "traffic_r06.json")
KERNEL_SOURCES = ["balm_kernels.hip", "balm_math.h", "lvba_internal.h", "lvba_api.hip", "block_system.hip", "pair_lists.hip",
                  "host_tables.h"]


def kernel_source_sha16():
    """hash of the sources that decide what the evaluation kernels do and how they are launched: a committed PMC summary is
    bound to it (a `git` hash would never match: the summary is committed AFTER the run it came from)"""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "global-lvba_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def read_traffic(which, config, world, kernels):
    """HBM bytes per launch from the committed PMC passes (tools/gpu_pmc2.sh -> tools/make_traffic.py -> profiles/), or null.
    PMC counters cannot be collected inside a timed run (rocprofv3 wraps the process), so the figure comes from a separate
    profiled run of the same command; it is REFUSED (null) unless that run was of THIS configuration on one GPU, profiled the
    kernels timed here, and was built from the kernel sources that are in the tree now."""
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as f:
            t = json.load(f)
        if t.get("config") != config or world != 1 or t.get(which + "_kernels") != kernels:
            return None
        if t.get("kernel_source_sha16") != kernel_source_sha16():
            return None
        return t.get(which)
    except Exception:
        return None


if __name__ == "__main__":
    main()
