#!/bin/bash
# the whole GPU suite with the fp32 Y records switched on: which parity tests notice?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4t; mkdir -p $O; cd $R
LVBA_Y32=1 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 | tee $O/gpu_tests_y32.txt
exit 0
