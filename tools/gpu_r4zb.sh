#!/bin/bash
# kernel trace of the front-end legs (upload, voxel map, window stage) after the late round-4 changes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4zb; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pfe -o stats -- python $R/tools/window_leg_probe.py > $O/probe_under_rocprof.txt 2>&1
python $R/tools/rocpd_stats.py /tmp/pfe/stats_results.db $O/front_end_kernel_stats.csv > /dev/null
head -30 $O/front_end_kernel_stats.csv | cut -c1-170
exit 0
