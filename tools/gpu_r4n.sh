#!/bin/bash
# round 4, late: 128 x 128 bulk tiles with operand chunks loaded straight into LDS (LVBA_BULK_TILE=sq)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4n; mkdir -p $O
cd $R/tools
for v in "-DLVBA_MB_DB=3" "-DLVBA_MB_DB=3 -DLVBA_SQ_KC=8" "-DLVBA_MB_DB=0"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value $v solver_microbench.hip -o /tmp/smb 2>&1 | grep -E "error"
  echo "=== microbench $v"; timeout 120 /tmp/smb 12000 2597 | grep -E "look-ahead|2 problems|job alone|bulk tile|first .* workgroups|SAME" | cut -c1-420
done 2>&1 | tee $O/smb.txt
cd $R
echo "--- tests"
timeout 900 python -m pytest tests/test_gpu_balm.py -q -x -p no:cacheprovider -k "schedules" 2>&1 | tail -3
echo "--- headline leg"
for e in "LVBA_BULK_TILE=sq" "LVBA_X=0"; do
  timeout 600 env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-visual --no-front-end --no-y32 > $O/b.log 2>&1
  echo "$e: $(grep -o '"ms_per_step": [0-9.]*' $O/b.log | head -1) $(grep -o '"stage_ms": {[^}]*}' $O/b.log) $(grep -o '"parity": {[^}]*}' $O/b.log | head -1)"
done
exit 0
