#!/bin/bash
# round 4: grouped handles on plain pair lists (window stage), the multi-share window test, front-end numbers of the bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4j; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_window.py tests/test_gpu_balm.py -q -x -p no:cacheprovider -k "window or shares or grouped or lock_step or lidar" 2>&1 | tail -6
echo "=== window_bench 320 frames"
LVBA_TIMING=1 timeout 300 python tools/window_bench.py 320 100000 20 1 > $O/window_bench.json 2> $O/w.err
grep -E "window_ba\]|LM\] (set-up|refine|create|handle)|bs_build\] (pairs|upload|ordering)|balm_create\] checks" $O/w.err | tail -12 | tr '\n' ';'; echo; grep -o '"gpu_s_per_window": [0-9.e-]*' $O/window_bench.json
echo "=== bench front end"
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-visual --no-y32 > $O/b.json 2> $O/b.err
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r4j/b.json') if l.startswith('{')][-1])
print({k:v for k,v in d.get('front_end',{}).items() if k in ('upload_ms','map_ms','points_per_s','points_per_s_end_to_end','window_ba')})
PY
exit 0
