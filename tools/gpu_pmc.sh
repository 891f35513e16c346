#!/bin/bash
# one PMC pass over a short bench run; prints per-kernel mean counter values.  usage: gpu_pmc.sh <COUNTERS...>
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/pmc1
rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc1 -o p -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-visual > /dev/null 2>&1
python $R/tools/rocpd_pmc.py /tmp/pmc1/p_results.db | grep -E "balm_|Name" | cut -c1-140
