#!/bin/bash
# quick check: BALM parity tests + stage times of the headline leg (optionally under several env settings given as args)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/quick; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_balm.py tests/test_gpu_multirank.py -q -x -p no:cacheprovider 2>&1 | tail -4
i=0
for e in "${@:-LVBA_X=0}"; do
  i=$((i+1))
  env $e python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-visual --no-front-end --no-y32 > $O/b_$i.log 2>&1
  echo "$e: $(grep -o '"ms_per_step": [0-9.]*' $O/b_$i.log | head -1) $(grep -o '"stage_ms": {[^}]*}' $O/b_$i.log)"
done
