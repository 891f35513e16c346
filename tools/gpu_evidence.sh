#!/bin/bash
# The round's evidence run (gpurun -- 'bash tools/gpu_evidence.sh <git hash> [round tag, default r05]'): (1) the whole GPU suite + smoke, (2) default bench.py, (3) rocprofv3 --kernel-trace --stats of the headline
# leg, (4) PMC passes of the headline leg: FETCH_SIZE / WRITE_SIZE (-> traffic json; also with LVBA_Y32=1) and the matrix-pipe
# counters of the solver kernels, (5) C2 and C4 with their parity legs, (6) visual stage, window stage.  Only summaries return.
TAG=${2:-r06}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | tee $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee -a $O/gpu_tests.txt
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.json; echo
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pb1 -o stats -- python $R/bench.py --no-cpu-baseline --no-visual --no-front-end --no-y32 > $O/bench_headline_under_rocprof.json 2>&1
python $R/tools/rocpd_stats.py /tmp/pb1/stats_results.db $O/headline_kernel_stats.csv > /dev/null
head -16 $O/headline_kernel_stats.csv | cut -c1-150
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES"; do
  t=$(echo $c | cut -d' ' -f1); rm -rf /tmp/pmc_$t
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$t -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-visual --no-front-end --no-y32 > $O/bench_pmc_$t.log 2>&1
  python $R/tools/rocpd_pmc.py /tmp/pmc_$t/p_results.db /tmp/pmc_$t.csv > /dev/null
  grep -E "Name,Counter|balm_|ldlt_|reduce_chunks|retract|predicted" /tmp/pmc_$t.csv > $O/pmc_$t.csv
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcy_$c
  LVBA_Y32=1 timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmcy_$c -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-visual --no-front-end --no-y32 > $O/bench_pmc_y32_$c.log 2>&1
  python $R/tools/rocpd_pmc.py /tmp/pmcy_$c/p_results.db /tmp/pmcy_$c.csv > /dev/null
  grep -E "Name,Counter|balm_" /tmp/pmcy_$c.csv > $O/pmc_y32_$c.csv
done
cd $R
python tools/make_traffic.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv $O/traffic.json C3 ${1:-unknown}
python tools/make_traffic.py $O/pmc_y32_FETCH_SIZE.csv $O/pmc_y32_WRITE_SIZE.csv $O/traffic_y32.json C3 ${1:-unknown}
grep -E "ldlt_step|ldlt_diag" $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES.csv | sed 's/(.*)"/"/' | cut -c1-160
timeout 600 python bench.py --config C2 --no-visual --no-front-end --no-reference-baseline > $O/bench_c2.json 2> $O/bench_c2.err
timeout 900 python bench.py --config C4 --steps 10 --warmup 2 --no-visual --no-front-end --no-reference-baseline --no-y32 > $O/bench_c4_1gpu.json 2> $O/bench_c4.err
tail -c 400 $O/bench_c2.json; echo; tail -c 400 $O/bench_c4_1gpu.json; echo; tail -3 $O/bench_c4.err
LVBA_TIMING=vis timeout 300 python tools/visual_bench.py 2000 5 > $O/visual_bench.json 2> $O/visual_bench.err; tail -c 300 $O/visual_bench.json
timeout 300 python tools/window_bench.py 320 100000 20 1 > $O/window_bench.json 2> $O/window_bench.err; tail -c 200 $O/window_bench.json; echo
# the solver on graphs that are not the BASELINE ring (band / nested dissection), and a kernel trace of a dissected solve
timeout 900 python tools/graph_bench.py > $O/graph_bench.json 2> $O/graph_bench.err; tail -c 300 $O/graph_bench.json; echo
cd /tmp; rm -rf /tmp/ndp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ndp -o nd -- python $R/tools/nd_probe.py 10000 3 > $O/nd_probe_under_rocprof.json 2>&1
python $R/tools/rocpd_stats.py /tmp/ndp/nd_results.db $O/nd_kernel_stats.csv > /dev/null; grep -E "step2|nd_|ldlt_fwd" $O/nd_kernel_stats.csv | cut -c1-70,120-190
cd $R
