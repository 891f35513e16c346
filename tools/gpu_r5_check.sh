#!/bin/bash
# round 5: the whole GPU suite + smoke + the default bench line (the driver's own round-end sequence), outputs under gpurun_out/<tag>
TAG=${1:-r5_check}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6 | tee $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee -a $O/gpu_tests.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - "$O/bench_default.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'stage_ms')})
print('timing', d['timing']['ms_per_step_min'], d['timing']['ms_per_step_max'])
print('roofline', d['roofline']['frac'], d['roofline']['traffic'], 'solve', d['roofline_solve']['frac'], 'cost', d['roofline_cost_pass_in_loop']['frac'])
print('parity ok', d.get('parity', {}).get('ok'))
fe = d.get('front_end', {})
print('front_end', fe.get('map_ms'), fe.get('roofline', {}).get('frac'), fe.get('window_ba', {}).get('ms_per_window'), fe.get('window_ba', {}).get('stage_ms_per_window'), fe.get('error'))
PY
exit 0
