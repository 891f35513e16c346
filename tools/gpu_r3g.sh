#!/bin/bash
# round 3: visual-stage tests, then tools/visual_bench.py (in-loop LM iteration time) under each given switch, same box
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_visual.py -q -x -p no:cacheprovider 2>&1 | tail -8
for e in "${@:-LVBA_X=0}"; do
  timeout 600 env $e python tools/visual_bench.py 2000 5 > /tmp/vb.log 2>&1
  echo "$e: $(tail -1 /tmp/vb.log | grep -o '"cap_50.*')"
  grep "visual profile" /tmp/vb.log | tail -2
done
exit 0
