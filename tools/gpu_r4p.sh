#!/bin/bash
# stream / pinned caches, trusted create, per-thread upload slots: the window leg and the upload again, then the tests that touch them
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4p; mkdir -p $O; cd $R
LVBA_TIMING=1 timeout 600 python tools/window_leg_probe.py > $O/probe.txt 2>&1
grep -E "^upload|^window_ba|^---|^\[window_ba\]" $O/probe.txt | tail -40
for t in 1 2 4 8 16; do echo "threads $t: $(LVBA_UPLOAD_THREADS=$t timeout 300 python tools/window_leg_probe.py 2>&1 | grep '^upload' | tr '\n' ' ')"; done
timeout 1200 python -m pytest tests/test_gpu_window.py tests/test_gpu_voxel.py tests/test_gpu_fusion.py tests/test_gpu_pipeline.py -q -x -p no:cacheprovider 2>&1 | tail -4
exit 0
