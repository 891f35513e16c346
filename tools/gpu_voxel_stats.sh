#!/bin/bash
# rocprofv3 kernel-trace stats of the voxel front-end bench; summary CSV only.  usage: gpu_voxel_stats.sh <tag> [frames] [pts]
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/pv_$tag -o stats -- python $R/tools/voxel_bench.py "$@" > $O/voxel_bench.log 2>&1
python $R/tools/rocpd_stats.py /tmp/pv_$tag/stats_results.db $O/kernel_stats.csv > /dev/null
cd $R
head -25 $O/kernel_stats.csv | cut -c1-170
tail -2 $O/voxel_bench.log | cut -c1-1500
