#!/bin/bash
# rocprofv3 kernel-trace stats of the probe.  usage: gpu_prof.sh <tag> <probe args...>
tag=$1; shift
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o $tag -- python $GRAFT_REPO_ROOT/tools/gpu_probe.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag/probe.log 2>&1
echo "exit $?" >> $GRAFT_REPO_ROOT/gpurun_out/$tag/probe.log
cd $GRAFT_REPO_ROOT
find gpurun_out/$tag -name "*stats*" | head; 
f=$(find gpurun_out/$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 "$f"
find gpurun_out/$tag -name "*kernel_trace.csv" -size +20M -delete
tail -25 gpurun_out/$tag/probe.log
