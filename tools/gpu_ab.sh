#!/bin/bash
# A/B of library builds / switches on ONE box: headline stage times.  usage: gpu_ab.sh "<env assignments>" ...   (LVBA_HIP_LIB=ab/x.so picks a build)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O
cd $R
i=0
for e in "$@"; do
  i=$((i+1))
  env $e python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-visual --no-front-end --no-y32 > $O/b_$i.json 2> $O/b_$i.err
  echo "$e: $(grep -o '"ms_per_step": [0-9.]*' $O/b_$i.json | head -1) $(grep -o '"stage_ms": {[^}]*}' $O/b_$i.json)"
done
