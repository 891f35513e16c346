#!/bin/bash
# what a flag-driven (launch-free) form would add per bulk tile: fences around every tile of the two-problem job
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4r; mkdir -p $O
cd $R/tools
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -DLVBA_MB_DB=0 solver_microbench.hip -o /tmp/smb 2>&1 | grep -E "error"
timeout 120 /tmp/smb 12000 2597 | grep -E "2 problems" | cut -c1-300 | tee $O/smb.txt
exit 0
