#!/bin/bash
# kernel trace of the window stage (tools/window_bench.py, 16 windows of 20 x 100 k points), batched LM (default)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/winprof; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/wp -o stats -- python $R/tools/window_bench.py 320 100000 20 0 > $O/bench.log 2>&1
python $R/tools/rocpd_stats.py /tmp/wp/stats_results.db $O/kernel_stats.csv > /dev/null
head -40 $O/kernel_stats.csv | cut -c1-140
