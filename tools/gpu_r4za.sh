#!/bin/bash
# issue priority of the row workgroups (they, not the bulk tiles, are what a two-ended launch waits for?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4za; mkdir -p $O; cd $R
for e in "LVBA_ROW_PRIO=0" "LVBA_ROW_PRIO=1" "LVBA_ROW_PRIO=2" "LVBA_ROW_PRIO=3" "LVBA_ROW_PRIO=0" "LVBA_ROW_PRIO=2"; do
  timeout 600 env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-visual --no-front-end --no-y32 > $O/b.log 2>&1
  echo "$e: $(grep -o '"ms_per_step": [0-9.]*' $O/b.log | head -1) $(grep -o '"solve": [0-9.]*' $O/b.log | head -1) $(grep -o '"last_cost": [0-9.e-]*' $O/b.log | head -1)"
done 2>&1 | tee $O/sweep.txt
timeout 900 python -m pytest tests/test_gpu_balm.py -q -x -p no:cacheprovider -k "solve_matches or refine_trace" 2>&1 | tail -2
exit 0
