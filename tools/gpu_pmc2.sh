#!/bin/bash
# two separate PMC passes (FETCH_SIZE, WRITE_SIZE) over a short headline-only bench run; CSV summaries into gpurun_out/pmc/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-visual --no-front-end --no-y32 > $O/bench_$c.log 2>&1
  python $R/tools/rocpd_pmc.py /tmp/pmc_$c/p_results.db $O/pmc_$c.csv > /dev/null
done
python $R/tools/make_traffic.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv $O/traffic.json; grep -E "balm_" $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv | cut -c1-170
