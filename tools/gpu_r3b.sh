#!/bin/bash
# round 3: solver microbenchmark (sweep vs blocked diagonal), solver tests, then the headline leg under the given switches
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b; mkdir -p $O
cd $R
bash tools/gpu_smb.sh 12000 2597 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_balm.py -q -x -p no:cacheprovider 2>&1 | tail -8
i=0
for e in "${@:-LVBA_X=0}"; do
  i=$((i+1))
  timeout 600 env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-visual --no-front-end > $O/b_$i.log 2>&1
  echo "$e: $(grep -o '"ms_per_step": [0-9.]*' $O/b_$i.log) $(grep -o '"stage_ms": {[^}]*}' $O/b_$i.log) $(grep -o '"avg_ms": [0-9.]*' $O/b_$i.log | head -3 | tr '\n' ' ')"
  tail -3 $O/b_$i.log | cut -c1-300 | grep -i "error\|Traceback"
done
