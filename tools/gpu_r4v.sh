#!/bin/bash
# panel q's share of the next diagonal block formed off the chain (row 1 of the launch before)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4v; mkdir -p $O
cd $R/tools
for v in "-DLVBA_MB_DB=0" "-DLVBA_MB_DB=0 -DLVBA_MB_NODQ"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value $v solver_microbench.hip -o /tmp/smb 2>&1 | grep -E "error"
  echo "=== microbench $v"; timeout 120 /tmp/smb 12000 2597 | grep -E "look-ahead|2 problems: roles|phases" | cut -c1-420
done 2>&1 | tee $O/smb.txt
cd $R
echo "--- tests"
timeout 1200 python -m pytest tests/test_gpu_balm.py tests/test_gpu_visual.py -q -x -p no:cacheprovider -k "solve or schedules or cyclic or refine_trace or c3_scale" 2>&1 | tail -3
echo "--- headline leg"
for e in "LVBA_X=0" "LVBA_CHAIN_DQ=0"; do
  timeout 600 env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-visual --no-front-end --no-y32 > $O/b.log 2>&1
  echo "$e: $(grep -o '"ms_per_step": [0-9.]*' $O/b.log | head -1) $(grep -o '"stage_ms": {[^}]*}' $O/b.log)"
done
exit 0
