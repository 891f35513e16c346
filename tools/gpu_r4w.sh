#!/bin/bash
# per-launch durations of the solve with panel q's share off / on the chain
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4w; mkdir -p $O; cd $R
export TMPDIR=/tmp
for m in 1 0 2; do
  (cd /tmp && LVBA_CHAIN_DQ=$m timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pb_$m -o stats -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-visual --no-front-end --no-y32 > $O/bench_prof_$m.log 2>&1
   python $R/tools/rocpd_timeline.py /tmp/pb_$m/stats_results.db $O/timeline_$m.csv 1200 > /dev/null)
  LVBA_CHAIN_DQ=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-visual --no-front-end --no-y32 > $O/b.log 2>&1
  echo "LVBA_CHAIN_DQ=$m: $(grep -o '"ms_per_step": [0-9.]*' $O/b.log | head -1) $(grep -o '"stage_ms": {[^}]*}' $O/b.log)"
done
exit 0
