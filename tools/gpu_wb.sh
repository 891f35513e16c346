#!/bin/bash
# window stage, joint passes against per-window passes: tools/window_bench.py at three shapes, stage times to stderr
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wb; mkdir -p $O; cd $R
for shape in "320 100000 20" "400 40000 10" "64 250000 16"; do
  for j in 1 0; do
    echo "== $shape joint=$j"
    LVBA_WINDOW_JOINT_MAP=$j LVBA_TIMING=1 timeout 600 python tools/window_bench.py $shape 0 2> $O/err_${j}.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('gpu_s','gpu_s_per_window')}, {k: d['windows'][0][k] for k in ('map_ms','setup_ms','solve_ms','merge_ms','n_anchor_points')})"
    grep "window_ba\]\|window_ba merge" $O/err_${j}.txt | tail -14
  done
done
