#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_balm.py tests/test_gpu_voxel.py -q -x -p no:cacheprovider -k "y32 or eval_blocks_is or strided_scans" 2>&1 | tail -15
exit 0
