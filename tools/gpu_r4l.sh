#!/bin/bash
# rehearsal of the multi-rank bench line with the window leg on (two ranks on the one GPU, gloo transport)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4l; mkdir -p $O; cd $R
timeout 900 python bench.py --gpus 2 --transport gloo --same-device --steps 3 --warmup 1 --config C2 --no-cpu-baseline --no-visual > $O/b2.json 2> $O/b2.err
tail -5 $O/b2.err | cut -c1-300
python - <<'PY'
import json
d=json.loads([l for l in open('/root/repo/gpurun_out/r4l/b2.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','n_gpus','ms_per_step','scaling')}); print(d['config']['sharding']); print(json.dumps(d.get('window_stage_all_ranks'))[:600]); print(json.dumps(d['scaling_model'])[:700])
PY
exit 0
