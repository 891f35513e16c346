#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/nd2; mkdir -p $O; cd $R
python tools/nd_probe.py 2000 6 2>/dev/null | tail -1
python tools/nd_probe.py 10000 3 2>/dev/null | tail -1
timeout 900 python -m pytest tests/test_gpu_nd.py tests/test_gpu_multirank.py -x -q -p no:cacheprovider 2>&1 | tail -3
