#!/bin/bash
# voxel-map and anchor-leaf sorts on the bits that vary
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4q; mkdir -p $O; cd $R
for e in "LVBA_SORT_BITS=varying" "LVBA_SORT_BITS=full"; do
  echo "=== $e"
  env $e LVBA_TIMING=1 timeout 600 python tools/window_leg_probe.py > $O/probe_${e#*=}.txt 2>&1
  grep -E "^voxel map|^window_ba|^\[window_ba\]" $O/probe_${e#*=}.txt | tail -11
done
timeout 1200 python -m pytest tests/test_gpu_window.py tests/test_gpu_voxel.py tests/test_gpu_fusion.py tests/test_gpu_pipeline.py -q -x -p no:cacheprovider 2>&1 | tail -4
exit 0
