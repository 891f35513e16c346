#!/bin/bash
# joint voxel map of all windows: tests, then the window leg and window_bench with and without it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4u; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_voxel.py tests/test_gpu_window.py tests/test_gpu_pipeline.py -q -x -p no:cacheprovider 2>&1 | tail -6
for e in "LVBA_WINDOW_JOINT_MAP=1" "LVBA_WINDOW_JOINT_MAP=0"; do
  echo "=== $e"
  env $e LVBA_TIMING=1 timeout 600 python tools/window_leg_probe.py > $O/probe_${e#*=}.txt 2>&1
  grep -E "^window_ba|^\[window_ba\]" $O/probe_${e#*=}.txt | tail -5
  env $e timeout 300 python tools/window_bench.py 320 100000 20 1 2> $O/wb_${e#*=}.err | tail -c 400; echo
done
exit 0
