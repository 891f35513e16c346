#!/bin/bash
# PMC passes over the visual stage alone (tools/visual_bench.py; kernel-trace + counters only), one pass per quoted counter list.
# usage: gpu_vispmc.sh "<counters pass 1>" ["<counters pass 2>" ...]; summaries -> gpurun_out/vispmc/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/vispmc; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
i=0
for c in "$@"; do
  i=$((i+1)); rm -rf /tmp/vp_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/vp_$i -o p -- python $R/tools/visual_bench.py 2000 1 > $O/pass_$i.log 2>&1
  python $R/tools/rocpd_pmc.py /tmp/vp_$i/p_results.db $O/pass_$i.csv > /dev/null
  grep -E "vis_cam_kernel|vis_point_kernel|vis_back_kernel|bcr_level" $O/pass_$i.csv | sed 's/(.*)"/"/' | cut -c1-160
done
