#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/nd2; mkdir -p $O; cd $R
export TMPDIR=/tmp
python tools/nd_probe.py 2000 6 2>/dev/null | tail -1
python tools/nd_probe.py 10000 3 2>/dev/null | tail -1
timeout 900 python -m pytest tests/test_gpu_nd.py -x -q -p no:cacheprovider 2>&1 | tail -3
cd /tmp; rm -rf /tmp/ndp
rocprofv3 --kernel-trace -d /tmp/ndp -o nd -- python $R/tools/nd_probe.py 2000 2 > $O/prof.log 2>&1
python $R/tools/rocpd_timeline.py /tmp/ndp/nd_results.db $O/timeline_threads.csv 700
