#!/bin/bash
# round 4, fourth GPU call: chain workgroups alone on their CUs; phases of the chain role; the whole GPU suite; the default bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4d; mkdir -p $O
cd $R/tools
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value solver_microbench.hip -o /tmp/smb 2>&1 | grep -E "error"
timeout 120 /tmp/smb 12000 2597 | grep -E "look-ahead|2 problems|job alone|K1 blocked|chain block|phases|return at once" | cut -c1-400 | tee $O/smb.txt
cd $R
echo "--- headline leg"
for e in "LVBA_X=0" "LVBA_CHAIN_ALONE=0" "LVBA_SOLVER=r3"; do
  timeout 600 env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-visual --no-front-end --no-y32 > $O/b.log 2>&1
  echo "$e: $(grep -o '"ms_per_step": [0-9.]*' $O/b.log | head -1) $(grep -o '"stage_ms": {[^}]*}' $O/b.log)"
done
echo "--- whole GPU suite"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8
echo "--- default bench"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r4d/bench.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','stage_ms')}); print('roofline frac', d['roofline']['frac'], 'y32', d.get('y32_mode')); print('front_end', {k:v for k,v in d.get('front_end',{}).items() if k in ('upload_ms','map_ms','points_per_s','points_per_s_end_to_end','window_ba')}); print('parity', d.get('parity'))
PY
echo "--- window stage, 16 windows"
LVBA_TIMING=1 timeout 300 python tools/window_bench.py 320 > $O/window.json 2> $O/window.err; grep -E "window_ba|bs_build|balm_create|finalize" $O/window.err | tail -28; cut -c1-200 $O/window.json
exit 0
