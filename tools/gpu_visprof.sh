#!/bin/bash
# visual stage: in-loop iteration time (tools/visual_bench.py), then the same under rocprofv3 --kernel-trace: per-kernel stats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/visprof; rm -rf $O; mkdir -p $O
cd $R
for e in "${@:-LVBA_X=0}"; do
  echo "$e: $(timeout 600 env $e python tools/visual_bench.py 2000 5 2>&1 | tail -1)"
done
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pb_vis -o stats -- python $R/tools/visual_bench.py 2000 1 > $O/under_rocprof.log 2>&1
python $R/tools/rocpd_stats.py /tmp/pb_vis/stats_results.db $O/kernel_stats.csv > /dev/null
python $R/tools/rocpd_timeline.py /tmp/pb_vis/stats_results.db $O/timeline.csv 600
cd $R
tail -1 $O/under_rocprof.log
grep "lvba\|rocclr" $O/kernel_stats.csv | cut -c1-110,200-260 | head -40
exit 0
