#!/bin/bash
# issue priority of the bulk tiles, row 1 alone on its CU, panel q's share off the chain: the solve under the combinations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4x; mkdir -p $O; cd $R
for e in "LVBA_CHAIN_DQ=0 LVBA_BULK_PRIO=0" "LVBA_CHAIN_DQ=0 LVBA_BULK_PRIO=1" "LVBA_CHAIN_DQ=0 LVBA_BULK_PRIO=2" \
         "LVBA_CHAIN_DQ=1 LVBA_BULK_PRIO=1" "LVBA_CHAIN_DQ=1 LVBA_BULK_PRIO=1 LVBA_ROW1_ALONE=1" "LVBA_CHAIN_DQ=1 LVBA_BULK_PRIO=0 LVBA_ROW1_ALONE=1" \
         "LVBA_CHAIN_DQ=2 LVBA_BULK_PRIO=1" "LVBA_CHAIN_DQ=1 LVBA_BULK_PRIO=2 LVBA_ROW1_ALONE=1"; do
  timeout 600 env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-visual --no-front-end --no-y32 > $O/b.log 2>&1
  echo "$e: $(grep -o '"ms_per_step": [0-9.]*' $O/b.log | head -1) $(grep -o '"solve": [0-9.]*' $O/b.log | head -1) $(grep -o '"ok": [a-z]*' $O/b.log | head -1)"
done 2>&1 | tee $O/sweep.txt
exit 0
