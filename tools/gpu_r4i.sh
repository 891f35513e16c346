#!/bin/bash
# round 4: window stage set-up experiments, the multi-share window test, C4's worst block
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4i; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_window.py -q -x -p no:cacheprovider -k "shares" 2>&1 | tail -25
for e in "LVBA_X=0" "LVBA_PAIR_SORT=0" "LVBA_PAIR_WINDOW=0"; do
  echo "=== window_bench 320 frames, $e"
  LVBA_TIMING=1 timeout 300 env $e python tools/window_bench.py 320 100000 20 0 > $O/w.json 2> $O/w.err
  grep -E "window_ba\]|LM\] (set-up|refine|create|handle)|bs_build\] (pairs|upload|ordering)" $O/w.err | tail -11 | tr '\n' ';'; echo; grep -o '"gpu_s_per_window": [0-9.e-]*' $O/w.json
done
echo "--- C4 worst block"
timeout 900 python tools/worst_block.py C4 2>&1 | tail -2 | tee $O/worst_block_c4.json
exit 0
