#!/bin/bash
# round 3: the config-size parity tests with full failure output, the BALM tests, then the headline leg under the given switches
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_config_parity.py::test_c4_cost_and_one_lm_iteration tests/test_gpu_multirank.py::test_c3_two_ranks_full_lm_trace -q -p no:cacheprovider 2>&1 | grep -v "^$" | tail -60 | cut -c1-220
timeout 900 python -m pytest tests/test_gpu_balm.py tests/test_gpu_window.py -q -x -p no:cacheprovider 2>&1 | tail -6
i=0
for e in "${@:-LVBA_X=0}"; do
  i=$((i+1))
  timeout 600 env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-visual --no-front-end > $O/b_$i.log 2>&1
  echo "$e: $(grep -o '"ms_per_step": [0-9.]*' $O/b_$i.log) $(grep -o '"stage_ms": {[^}]*}' $O/b_$i.log) $(grep -o '"avg_ms": [0-9.]*' $O/b_$i.log | head -3 | tr '\n' ' ')"
  tail -3 $O/b_$i.log | cut -c1-300 | grep -i "error\|Traceback"
done
