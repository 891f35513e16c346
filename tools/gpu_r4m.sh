#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_window.py -q -x -p no:cacheprovider 2>&1 | tail -5
exit 0
