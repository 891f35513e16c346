#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4y; mkdir -p $O; cd $R
for e in "LVBA_CHAIN_DQ=0" "LVBA_CHAIN_DQ=3" "LVBA_CHAIN_DQ=3 LVBA_ROW1_ALONE=1" "LVBA_CHAIN_DQ=0" "LVBA_CHAIN_DQ=3 LVBA_ROW1_ALONE=1"; do
  timeout 600 env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-visual --no-front-end --no-y32 > $O/b.log 2>&1
  echo "$e: $(grep -o '"ms_per_step": [0-9.]*' $O/b.log | head -1) $(grep -o '"solve": [0-9.]*' $O/b.log | head -1) $(grep -o '"ok": [a-z]*' $O/b.log | head -1)"
done 2>&1 | tee $O/sweep.txt
timeout 1200 python -m pytest tests/test_gpu_balm.py -q -x -p no:cacheprovider -k "schedules" 2>&1 | tail -3
exit 0
