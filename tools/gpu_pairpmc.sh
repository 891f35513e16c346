#!/bin/bash
# PMC passes over the headline leg for A/B variants selected through the environment.  usage: gpu_pairpmc.sh tag "ENV=.. ENV=.." counters...
tag=$1; envs=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pairpmc; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
i=0
for c in "$@"; do
  i=$((i+1)); rm -rf /tmp/pp_$tag_$i
  env $envs rocprofv3 --kernel-trace --pmc $c -d /tmp/pp_${tag}_$i -o p -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-visual --no-front-end > $O/${tag}_$i.log 2>&1
  python $R/tools/rocpd_pmc.py /tmp/pp_${tag}_$i/p_results.db $O/${tag}_$i.csv > /dev/null
  grep -E "balm_pair|balm_factor|balm_voxel" $O/${tag}_$i.csv | cut -c1-200
done
