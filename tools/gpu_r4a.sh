#!/bin/bash
# round 4, first GPU call: Ceres/Eigen probe, solver tests under the look-ahead schedule, the solver
# microbenchmark, the headline leg under the schedule switches, the whole GPU suite, eager vs graph.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4a; mkdir -p $O
cd $R
{ echo "== find / for Ceres / Eigen / SuiteSparse / glog (none = not installed on the GPU box)";
  find / -xdev \( -iname 'ceres*' -o -name 'Eigen' -o -iname 'libceres*' -o -iname 'eigen3*' -o -iname 'libcholmod*' -o -iname 'suitesparse*' -o -iname 'libglog*' \) -not -path '/proc/*' -not -path "$R/*" -not -path '/root/repo/*' 2>/dev/null | head -50;
  echo "== pkg-config"; pkg-config --list-all 2>/dev/null | grep -i -E 'ceres|eigen|glog|suitesparse' ; echo "== done"; } > $O/ceres_probe.txt 2>&1
echo "--- probe: $(wc -l < $O/ceres_probe.txt) lines"; cat $O/ceres_probe.txt | head -20
echo "--- solver tests"
timeout 900 python -m pytest tests/test_gpu_balm.py -q -x -p no:cacheprovider -k "solve or schedules or refine_trace or grouped" 2>&1 | tail -8
echo "--- microbench"
(cd tools && hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value solver_microbench.hip -o /tmp/smb 2>&1 | grep -E "error" ; timeout 120 /tmp/smb 12000 2597 | cut -c1-300) 2>&1 | tee $O/smb.txt
echo "--- headline leg under the switches"
i=0
for e in "LVBA_X=0" "LVBA_SOLVER=r3" "LVBA_RANK128=0" "LVBA_NO_GRAPH=1" "LVBA_SOLVER=r3 LVBA_NO_GRAPH=1"; do
  i=$((i+1))
  timeout 600 env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-visual --no-front-end > $O/b_$i.log 2>&1
  echo "$e: $(grep -o '"ms_per_step": [0-9.]*' $O/b_$i.log | head -1) $(grep -o '"stage_ms": {[^}]*}' $O/b_$i.log) $(grep -o '"last_cost": [0-9.e-]*' $O/b_$i.log)"
done
echo "--- whole GPU suite"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15
exit 0
