#!/bin/bash
# development run: the nested-dissection tests, the solver on graphs that are not the BASELINE ring (tools/graph_bench.py), and a
# kernel trace of the dissected solve on the parking-lot graph
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/nd; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_nd.py -x -q -p no:cacheprovider 2>&1 | tail -8 | tee $O/nd_tests.txt
timeout 1500 python tools/graph_bench.py --graphs lot ${1:-} > $O/graph_bench_lot.json 2> $O/graph_bench_lot.err; tail -3 $O/graph_bench_lot.err; cut -c1-1500 $O/graph_bench_lot.json | head -4
cd /tmp; rm -rf /tmp/ndp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ndp -o nd -- python $R/tools/graph_bench.py --graphs lot --sizes 2000 > $O/prof.log 2>&1
python $R/tools/rocpd_stats.py /tmp/ndp/nd_results.db $O/nd_kernel_stats.csv > /dev/null
head -14 $O/nd_kernel_stats.csv | cut -c1-130
