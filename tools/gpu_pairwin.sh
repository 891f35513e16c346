#!/bin/bash
# A/B of the window-grouped pair lists: eval stage time of the headline leg for several window sizes (voxels per window)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pairwin; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_balm.py tests/test_gpu_visual.py tests/test_gpu_window.py -q -x -p no:cacheprovider 2>&1 | tail -3
LVBA_PAIR_WINDOW=7 python -m pytest tests/test_gpu_balm.py -q -x -p no:cacheprovider 2>&1 | tail -3
for w in ${WINDOWS:-0 1024 2048 4096 8192 16384}; do
  LVBA_PAIR_WINDOW=$w python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-visual --no-front-end > $O/b_$w.log 2>&1
  echo "window $w: $(grep -o '"stage_ms": {[^}]*}' $O/b_$w.log) $(grep -o '"avg_ms": [0-9.]*' $O/b_$w.log | head -1)"
done
