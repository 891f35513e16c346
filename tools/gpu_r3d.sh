#!/bin/bash
# BALM tests, headline leg, one FETCH_SIZE pass (factor / pair kernels)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3d; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_balm.py tests/test_gpu_config_parity.py -q -x -p no:cacheprovider 2>&1 | tail -4
i=0
for e in "${@:-LVBA_X=0}"; do
  i=$((i+1))
  timeout 600 env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-visual --no-front-end > $O/b_$i.log 2>&1
  echo "$e: $(grep -o '"ms_per_step": [0-9.]*' $O/b_$i.log) $(grep -o '"stage_ms": {[^}]*}' $O/b_$i.log) $(grep -o '"avg_ms": [0-9.]*' $O/b_$i.log | head -3 | tr '\n' ' ')"
  tail -3 $O/b_$i.log | cut -c1-300 | grep -i "error\|Traceback"
done
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pf
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-visual --no-front-end > $O/pmc.log 2>&1
python $R/tools/rocpd_pmc.py /tmp/pf/p_results.db /tmp/pf.csv > /dev/null; grep -E "balm_" /tmp/pf.csv | sed 's/(.*)"/"/' | cut -c1-120
