#!/bin/bash
# round 4, fifth GPU call: where a bulk tile's time goes; timeline of the solve with the chain alone on its CU
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4e; mkdir -p $O
cd $R/tools
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value solver_microbench.hip -o /tmp/smb 2>&1 | grep -E "error"
timeout 120 /tmp/smb 12000 2597 | grep -E "2 problems|job alone|bulk tile|first .* workgroups|SAME" | cut -c1-400 | tee $O/smb.txt
cd $R
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pb_r4e -o stats -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-visual --no-front-end --no-y32 > $O/bench_prof.log 2>&1
 python $R/tools/rocpd_timeline.py /tmp/pb_r4e/stats_results.db $O/timeline.csv 1200 > /dev/null)
python - <<'PY'
import csv, statistics
rows=list(csv.DictReader(open('/root/repo/gpurun_out/r4e/timeline.csv')))
idx=[i for i,r in enumerate(rows) if r['name'].startswith('ldlt_prepare_band')]
seg=rows[idx[-2]:idx[-1]]
d=[float(r['dur_us']) for r in seg if r['name'].startswith('ldlt_step2')]
print(len(d),'step2 launches; sum %.0f us'%sum(d))
print('two-ended:', [round(x) for x in d[:76]])
print('S phase:', [round(x) for x in d[76:]])
PY
exit 0
