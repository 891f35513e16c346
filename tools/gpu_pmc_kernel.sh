#!/bin/bash
# PMC passes over the headline leg (kernel-trace + counters only), one pass per argument (a quoted counter list).
# usage: gpu_pmc_kernel.sh <tag> "<ENV=.. ENV=..>" "<counters pass 1>" ["<counters pass 2>" ...]; summaries -> gpurun_out/pmck/
tag=$1; envs=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmck; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
i=0
for c in "$@"; do
  i=$((i+1)); rm -rf /tmp/pk_${tag}_$i
  timeout 600 env $envs rocprofv3 --kernel-trace --pmc $c -d /tmp/pk_${tag}_$i -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-visual --no-front-end > $O/${tag}_$i.log 2>&1
  python $R/tools/rocpd_pmc.py /tmp/pk_${tag}_$i/p_results.db $O/${tag}_$i.csv > /dev/null
  grep -E "balm_|ldlt_step|ldlt_diag" $O/${tag}_$i.csv | sed 's/(.*)//' | cut -c1-160
done
