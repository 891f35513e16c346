#!/bin/bash
# final check of the round: the whole GPU suite, smoke, the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_final; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | tee $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee -a $O/gpu_tests.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r4_final/bench_default.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','stage_ms')}); print('roofline', d['roofline']['frac'], d['roofline']['traffic']); print('y32', d['y32_mode']['value'], d['y32_mode']['parity_vs_fp64_records']['ok']); print('parity ok', d['parity']['ok']); print('front_end', d['front_end']['upload_ms'], d['front_end']['window_ba']['ms_per_window'])
PY
exit 0
