#!/bin/bash
# kernel timeline of the last LM steps of the headline leg: step-launch durations in order (two-ended / S phase)
# usage: gpurun -- 'bash tools/gpu_trace.sh [tag] [env assignments...]'
TAG=${1:-trace}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/tr1
env "$@" timeout 600 rocprofv3 --kernel-trace -d /tmp/tr1 -o t -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-visual --no-front-end --no-y32 > $O/bench_under_trace.json 2> $O/bench_under_trace.err
python $R/tools/rocpd_timeline.py /tmp/tr1/t_results.db $O/timeline.csv 700 > /dev/null
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/timeline.csv")))
# last complete solve: from the last ldlt_prepare_band_kernel on
idx=[i for i,r in enumerate(rows) if r["name"].startswith("ldlt_prepare_band")]
if len(idx)>=2:
    seg=rows[idx[-2]:idx[-1]]
    steps=[float(r["dur_us"]) for r in seg if r["name"].startswith("ldlt_step2")]
    gaps=[float(r["gap_us"]) for r in seg if r["name"].startswith("ldlt_step2")]
    print("step launches:",len(steps),"sum %.1f us"%sum(steps),"gaps sum %.1f us"%sum(gaps))
    print("durations:"," ".join("%.1f"%s for s in steps))
    t0=float(seg[0]["start_us"]); t1=float(rows[idx[-1]]["start_us"])
    print("prepare_band -> next prepare_band: %.1f us"%(t1-t0))
    other={}
    for r in seg:
        if not r["name"].startswith("ldlt_step2"):
            other.setdefault(r["name"],[0,0.0]); other[r["name"]][0]+=1; other[r["name"]][1]+=float(r["dur_us"])
    for k,v in other.items(): print("  %-40s x%-3d %.1f us"%(k,v[0],v[1]))
    print("  gaps between all kernels of the segment: %.1f us"%sum(float(r["gap_us"]) for r in seg[1:]))
PY
