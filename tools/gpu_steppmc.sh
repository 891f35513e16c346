#!/bin/bash
# HBM / fabric traffic of the LDL^T update workgroups alone (tools/step_microbench in counter mode): FETCH_SIZE, WRITE_SIZE per launch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/steppmc; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for m in 0 1 2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/sp_$c
    rocprofv3 --kernel-trace --pmc $c -d /tmp/sp_$c -o p -- $R/tools/step_microbench 12000 2597 $m > $O/log_${m}_$c.txt 2>&1
    python $R/tools/rocpd_pmc.py /tmp/sp_$c/p_results.db $O/pmc_${m}_$c.csv > /dev/null
    echo "mode $m $c:"; grep -i "step_kernel" $O/pmc_${m}_$c.csv | cut -c1-60,150-400
  done
done
