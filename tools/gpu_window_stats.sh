#!/bin/bash
# rocprofv3 kernel-trace stats of the window-BA bench; summary CSV only.  usage: gpu_window_stats.sh <tag> [frames] [pts] [window]
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/pw_$tag -o stats -- python $R/tools/window_bench.py "$@" > $O/window_bench.log 2>&1
python $R/tools/rocpd_stats.py /tmp/pw_$tag/stats_results.db $O/kernel_stats.csv > /dev/null
cd $R
python - <<PY
import csv
rows=list(csv.reader(open("$O/kernel_stats.csv")))[1:]
tot=sum(float(r[2]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:28]:
    n=r[0]
    short=('rocprim:'+n.split('wrapped_')[1][:24]) if 'wrapped_' in n else n.split('(')[0][-44:]
    print(f"{short:46s} calls={r[1]:>5s} avg_us={float(r[3])/1e3:8.1f} total_ms={float(r[2])/1e6:7.2f}")
PY
