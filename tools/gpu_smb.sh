#!/bin/bash
# builds and runs tools/solver_microbench.hip on the GPU box.  usage: gpu_smb.sh [n] [bw]
cd $GRAFT_REPO_ROOT/tools
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value solver_microbench.hip -o /tmp/smb && /tmp/smb "$@" | cut -c1-400
