#!/bin/bash
# A/B of the two-ended band LDL^T: solver tests, then stage times of the headline leg and the visual leg with and without it
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/twist; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_balm.py tests/test_gpu_visual.py tests/test_gpu_config_parity.py -q -x -p no:cacheprovider 2>&1 | tail -4
for t in 0 1; do
  LVBA_TWIST=$t python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-front-end > $O/b_$t.log 2>&1
  echo "twist $t: $(grep -o '"ms_per_step": [0-9.]*' $O/b_$t.log) $(grep -o '"stage_ms": {[^}]*}' $O/b_$t.log) visual $(grep -o '"ms_per_iteration": [0-9.]*' $O/b_$t.log)"
done
