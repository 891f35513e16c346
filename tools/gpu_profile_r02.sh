#!/bin/bash
# Round-2 evidence run: (1) default bench.py, (2) rocprofv3 --kernel-trace --stats of the headline leg alone and of the full
# default command, (3) the two PMC passes (FETCH_SIZE / WRITE_SIZE) of the headline leg -> traffic json.  Only summaries return.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r02; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/pb1 -o stats -- python $R/bench.py --no-cpu-baseline --no-visual --no-front-end > $O/bench_headline_under_rocprof.json 2>&1
python $R/tools/rocpd_stats.py /tmp/pb1/stats_results.db $O/headline_kernel_stats.csv > /dev/null
rocprofv3 --kernel-trace --stats -d /tmp/pb2 -o stats -- python $R/bench.py --no-cpu-baseline > $O/bench_full_under_rocprof.json 2>&1
python $R/tools/rocpd_stats.py /tmp/pb2/stats_results.db $O/full_kernel_stats.csv > /dev/null
bash $R/tools/gpu_pmc2.sh > $O/pmc.log 2>&1
cp $R/gpurun_out/pmc/pmc_FETCH_SIZE.csv $R/gpurun_out/pmc/pmc_WRITE_SIZE.csv $R/gpurun_out/pmc/traffic.json $O/ 2>/dev/null
head -30 $O/headline_kernel_stats.csv | cut -c1-150
cat $O/traffic.json | head -5
# other configurations (headline leg only)
cd $R
python bench.py --config C2 --no-visual --no-front-end --no-reference-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --config C4 --steps 10 --warmup 2 --no-visual --no-front-end --no-cpu-baseline > $O/bench_c4_1gpu.json 2> $O/bench_c4.err
tail -c 600 $O/bench_c2.json; tail -c 400 $O/bench_c4_1gpu.json; tail -3 $O/bench_c4.err
# window stage (16 windows of 20 x 100 k points): batched LM (default) and one window at a time
for m in 1 0; do LVBA_WINDOW_BATCH=$m python tools/window_bench.py 320 100000 20 1 > $O/window_bench_batch$m.json 2> $O/window_bench_batch$m.err; tail -c 300 $O/window_bench_batch$m.json; echo; done
