#!/bin/bash
# development run: the nested-dissection tests, then the solver on graphs that are not the BASELINE ring (tools/graph_bench.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/nd; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_nd.py tests/test_abi.py -x -q -p no:cacheprovider 2>&1 | tail -15 | tee $O/nd_tests.txt
timeout 1500 python tools/graph_bench.py ${1:-} > $O/graph_bench.json 2> $O/graph_bench.err; tail -5 $O/graph_bench.err; cut -c1-1200 $O/graph_bench.json | head -12
