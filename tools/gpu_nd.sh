#!/bin/bash
# development run: the nested-dissection tests (+ optional extra pytest args)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/nd; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_nd.py -x -q -p no:cacheprovider "$@" 2>&1 | tail -40 | tee $O/nd_tests.txt
