#!/bin/bash
# builds and runs tools/bcr_microbench.hip on the GPU box.  usage: gpu_bcrmb.sh [n_cams] [band_blocks]
cd $GRAFT_REPO_ROOT/tools
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value bcr_microbench.hip -o /tmp/bcrmb && timeout 60 /tmp/bcrmb "$@" | cut -c1-300
