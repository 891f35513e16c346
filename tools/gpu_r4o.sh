#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4o; mkdir -p $O; cd $R
LVBA_TIMING=1 timeout 600 python tools/window_leg_probe.py > $O/probe.txt 2>&1
grep -v "^\[balm\|^\[bs" $O/probe.txt | tail -150
exit 0
