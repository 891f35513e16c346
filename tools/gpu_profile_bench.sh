#!/bin/bash
# rocprofv3 of the bench command: (1) kernel-trace stats, (2)/(3) separate PMC passes (FETCH_SIZE / WRITE_SIZE).
# The rocpd databases (60+ MB each) are summarised on the box and deleted; only CSV summaries come back.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_bench
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/pb -o stats -- python $R/bench.py --no-cpu-baseline > $O/bench_stats.log 2>&1
python $R/tools/rocpd_stats.py /tmp/pb/stats_results.db $O/kernel_stats.csv > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pb -o fetch -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_fetch.log 2>&1
python $R/tools/rocpd_pmc.py /tmp/pb/fetch_results.db $O/pmc_fetch.csv > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pb -o write -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_write.log 2>&1
python $R/tools/rocpd_pmc.py /tmp/pb/write_results.db $O/pmc_write.csv > /dev/null
cd $R
ls -la $O
grep lvba $O/kernel_stats.csv | cut -c1-160
grep lvba $O/pmc_fetch.csv | cut -c1-160
grep lvba $O/pmc_write.csv | cut -c1-160
