#!/bin/bash
# round 4, third GPU call: MFMA issue rate of one workgroup, the batched-broadcast pivot chain, who shares a CU with the chain
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c; mkdir -p $O
cd $R/tools
hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_rate.hip -o /tmp/mfma_rate 2>&1 | grep error; timeout 60 /tmp/mfma_rate 2>&1 | tee $O/mfma_rate.txt
for v in "" "-DLVBA_K1B_V1"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value $v solver_microbench.hip -o /tmp/smb 2>&1 | grep -E "error"
  echo "=== microbench $v"; timeout 120 /tmp/smb 12000 2597 | grep -E "look-ahead|2 problems|stagger|job alone|K1 blocked|chain block|distinct|return at once" | cut -c1-260
done 2>&1 | tee $O/smb.txt
cd $R
echo "--- tests"
timeout 900 python -m pytest tests/test_gpu_balm.py tests/test_gpu_visual.py -q -x -p no:cacheprovider -k "solve or schedules or cyclic or refine_trace" 2>&1 | tail -3
echo "--- headline leg"
for e in "LVBA_X=0" "LVBA_Y32=1"; do
  timeout 600 env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-visual --no-front-end > $O/b.log 2>&1
  echo "$e: $(grep -o '"ms_per_step": [0-9.]*' $O/b.log | head -1) $(grep -o '"stage_ms": {[^}]*}' $O/b.log)"
done
exit 0
