#!/bin/bash
# list the PMC counters the box offers (SQ / TA / TCP / TCC groups) -> gpurun_out/pmck/avail.txt, then run the given passes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmck; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -E "^\s*(Name|Counter_Name)?\s*:?\s*(SQ_|TA_|TCP_|TCC_|GRBM_)[A-Za-z0-9_]+" -o | sort -u > $O/avail.txt
wc -l $O/avail.txt
bash $R/tools/gpu_pmc_kernel.sh "$@"
