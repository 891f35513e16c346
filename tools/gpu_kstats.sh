#!/bin/bash
# average kernel durations (rocprofv3 --kernel-trace --stats) of the headline leg for several builds / switches
# usage: gpu_kstats.sh "<env assignments>" ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kstats; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
i=0
for e in "$@"; do
  i=$((i+1)); rm -rf /tmp/ks_$i
  e2=$(echo "$e" | sed "s#=ab/#=$R/ab/#"); env $e2 rocprofv3 --kernel-trace --stats -d /tmp/ks_$i -o s -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-visual --no-front-end --no-y32 > $O/bench_$i.json 2> $O/bench_$i.err
  python $R/tools/rocpd_stats.py /tmp/ks_$i/s_results.db $O/stats_$i.csv > /dev/null
  echo "== $e"
  python - <<PY
import csv
for r in csv.DictReader(open("$O/stats_$i.csv")):
    n=r["Name"]
    if any(k in n for k in ("balm_","ldlt_","reduce_chunks","retract","predicted")):
        print("  %-44s calls %5s avg %9.1f us"%(n.split("(")[0].replace("void ","").replace("lvba::","")[:44], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
