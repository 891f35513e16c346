#!/bin/bash
# A/B of library builds (LVBA_HIP_LIB) on the headline leg
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3e; mkdir -p $O
cd $R
i=0
for lib in "$@"; do
  i=$((i+1))
  timeout 600 env LVBA_HIP_LIB=$R/global-lvba_amd/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-visual --no-front-end > $O/b_$i.log 2>&1
  echo "$lib: $(grep -o '"ms_per_step": [0-9.]*' $O/b_$i.log) $(grep -o '"stage_ms": {[^}]*}' $O/b_$i.log) $(grep -o '"avg_ms": [0-9.]*' $O/b_$i.log | head -3 | tr '\n' ' ')"
  tail -3 $O/b_$i.log | cut -c1-300 | grep -i "error\|Traceback"
done
