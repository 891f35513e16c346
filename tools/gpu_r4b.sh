#!/bin/bash
# round 4, second GPU call: microbenchmark variants of the look-ahead launch, the narrow-band fix, timeline of the new solve,
# fp32-Y probe, window-stage timing breakdown
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4b; mkdir -p $O
cd $R/tools
for v in "" "-DLVBA_TP_PLAIN" "-DLVBA_BULK_NOSCHED"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value $v solver_microbench.hip -o /tmp/smb 2>&1 | grep -E "error"
  echo "=== microbench $v"; timeout 120 /tmp/smb 12000 2597 | grep -E "look-ahead|2 problems|stagger|job alone|K1 blocked" | cut -c1-200
done 2>&1 | tee $O/smb.txt
cd $R
echo "--- tests"
timeout 900 python -m pytest tests/test_gpu_balm.py tests/test_gpu_visual.py -q -x -p no:cacheprovider -k "solve or schedules or cyclic or c3_scale or refine_trace" 2>&1 | tail -5
echo "--- headline leg"
i=0
for e in "LVBA_X=0" "LVBA_SOLVER=r3"; do
  i=$((i+1))
  timeout 600 env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-visual --no-front-end > $O/b_$i.log 2>&1
  echo "$e: $(grep -o '"ms_per_step": [0-9.]*' $O/b_$i.log | head -1) $(grep -o '"stage_ms": {[^}]*}' $O/b_$i.log)"
done
echo "--- timeline of the new solve"
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pb_r4b -o stats -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-visual --no-front-end > $O/bench_prof.log 2>&1
 python $R/tools/rocpd_stats.py /tmp/pb_r4b/stats_results.db $O/kernel_stats.csv > /dev/null
 python $R/tools/rocpd_timeline.py /tmp/pb_r4b/stats_results.db $O/timeline.csv 1200 > /dev/null)
grep -E "ldlt|balm" $O/kernel_stats.csv | cut -c1-160
echo "--- fp32 Y probe"
timeout 600 python tools/y32_probe.py C2 C3 2>&1 | tail -4 | tee $O/y32.txt
echo "--- window stage"
LVBA_TIMING=1 timeout 300 python tools/window_bench.py > $O/window.json 2> $O/window.err; grep -E "window_ba|bs_build" $O/window.err | tail -40; cut -c1-300 $O/window.json
exit 0
