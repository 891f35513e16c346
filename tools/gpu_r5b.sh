#!/bin/bash
# ND at benchmark sizes: C4 line with the dissection model in its scaling_model; the graph bench (all graphs, both sizes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
timeout 900 python bench.py --config C4 --steps 10 --warmup 2 --no-visual --no-front-end --no-reference-baseline --no-y32 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
python - "$O/bench_c4.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'stage_ms')})
for n, v in d['scaling_model']['projected'].items(): print(n, round(v['ms_per_iteration'], 2), round(v['speedup'], 2), round(v['solve_ms'], 2), v['solve'][:110])
PY
timeout 1500 python tools/graph_bench.py > $O/graph_bench.json 2> $O/graph_bench.err; tail -2 $O/graph_bench.err
python - "$O/graph_bench.json" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{"graph"'):
        r = json.loads(l)
        w = r.get("without_dissection")
        print(r["graph"], r["n_poses"], r["default"]["path"][:60], "solve", round(r["default"]["solve_ms"], 2), "setup", round(r["default"]["setup_s"], 1), "res", r["default"]["residual_rel"], "| band", round(w["solve_ms"], 2) if w else None)
PY
