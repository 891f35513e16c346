#!/bin/bash
# round 3: solver tests (single problem + grouped windows), then the headline leg under each given switch, same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_balm.py tests/test_gpu_window.py -q -x -p no:cacheprovider 2>&1 | tail -8
i=0
for e in "${@:-LVBA_X=0}"; do
  i=$((i+1))
  timeout 600 env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-visual --no-front-end > $O/b_$i.log 2>&1
  echo "$e: $(grep -o '"ms_per_step": [0-9.]*' $O/b_$i.log) $(grep -o '"stage_ms": {[^}]*}' $O/b_$i.log) $(grep -o '"avg_ms": [0-9.]*' $O/b_$i.log | head -3 | tr '\n' ' ')"
done
exit 0
