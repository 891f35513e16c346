#!/bin/bash
# copy the summaries of tools/gpu_evidence.sh (gpurun_out/prof_<tag>/) into profiles/ under the round's names
T=${1:-r06}; S=gpurun_out/prof_$T; P=profiles
cp $S/bench_default.json $P/${T}_bench_c3.json
cp $S/bench_headline_under_rocprof.json $P/${T}_bench_c3_under_rocprof.json
cp $S/headline_kernel_stats.csv $P/${T}_bench_c3_kernel_stats.csv
cp $S/pmc_FETCH_SIZE.csv $P/${T}_bench_c3_pmc_fetch_size.csv
cp $S/pmc_WRITE_SIZE.csv $P/${T}_bench_c3_pmc_write_size.csv
cp $S/pmc_SQ_VALU_MFMA_BUSY_CYCLES.csv $P/${T}_bench_c3_pmc_mfma.csv
cp $S/pmc_y32_FETCH_SIZE.csv $P/${T}_bench_c3_y32_pmc_fetch_size.csv
cp $S/pmc_y32_WRITE_SIZE.csv $P/${T}_bench_c3_y32_pmc_write_size.csv
cp $S/traffic.json $P/traffic_$T.json
[ -f $S/traffic_y32.json ] && cp $S/traffic_y32.json $P/traffic_${T}_y32.json
cp $S/bench_c2.json $P/${T}_bench_c2.json
cp $S/bench_c4_1gpu.json $P/${T}_bench_c4_1gpu.json
cp $S/gpu_tests.txt $P/${T}_final_gpu_checks.txt
cp $S/graph_bench.json $P/${T}_graph_bench.json
cp $S/nd_kernel_stats.csv $P/${T}_nd_lot10000_kernel_stats.csv
cp $S/visual_bench.json $P/${T}_visual_bench.json
cp $S/window_bench.json $P/${T}_window_bench.json
ls -la $P | grep $T
