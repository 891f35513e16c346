#!/bin/bash
# rocprofv3 kernel-trace stats of a short bench run; summary CSV only.  usage: gpu_stats.sh <tag> [bench args]
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/pb_$tag -o stats -- python $R/bench.py --no-cpu-baseline "$@" > $O/bench.log 2>&1
python $R/tools/rocpd_stats.py /tmp/pb_$tag/stats_results.db $O/kernel_stats.csv > /dev/null
cd $R
grep lvba $O/kernel_stats.csv | cut -c1-150
grep -o '"stage_ms": {[^}]*}' $O/bench.log; grep -o '"ms_per_step": [0-9.]*' $O/bench.log
