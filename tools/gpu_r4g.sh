#!/bin/bash
# round 4, sixth GPU call: bulk tiles with two chunk buffers in LDS (one workgroup per CU); chain role trimmed
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4g; mkdir -p $O
cd $R/tools
for v in "-DLVBA_MB_DB=2" "-DLVBA_MB_DB=0"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value $v solver_microbench.hip -o /tmp/smb 2>&1 | grep -E "error"
  echo "=== microbench $v"; timeout 120 /tmp/smb 12000 2597 | grep -E "look-ahead|2 problems|job alone|bulk tile|first .* workgroups|SAME|phases" | cut -c1-420
done 2>&1 | tee $O/smb.txt
cd $R
echo "--- tests"
timeout 900 python -m pytest tests/test_gpu_balm.py tests/test_gpu_visual.py -q -x -p no:cacheprovider -k "solve or schedules or cyclic or refine_trace or c3_scale" 2>&1 | tail -3
echo "--- headline leg"
for e in "LVBA_X=0" "LVBA_BULK_TILE=k32" "LVBA_RANK128=0"; do
  timeout 600 env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-visual --no-front-end --no-y32 > $O/b.log 2>&1
  echo "$e: $(grep -o '"ms_per_step": [0-9.]*' $O/b.log | head -1) $(grep -o '"stage_ms": {[^}]*}' $O/b.log)"
done
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pb_r4g -o stats -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-visual --no-front-end --no-y32 > $O/bench_prof.log 2>&1
 python $R/tools/rocpd_timeline.py /tmp/pb_r4g/stats_results.db $O/timeline.csv 1200 > /dev/null)
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/r4g/timeline.csv')))
idx=[i for i,r in enumerate(rows) if r['name'].startswith('ldlt_prepare_band')]
seg=rows[idx[-2]:idx[-1]]
d=[float(r['dur_us']) for r in seg if r['name'].startswith('ldlt_step2')]
print(len(d),'step2 launches; sum %.0f us'%sum(d))
print('two-ended:', [round(x) for x in d[:76]])
print('S phase:', [round(x) for x in d[76:]])
PY
exit 0
