#!/bin/bash
cd $GRAFT_REPO_ROOT/tools
for v in 0 1 2 3 5; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -DLVBA_K1_VARIANT=$v solver_microbench.hip -o /tmp/smb$v && echo "variant $v: $(/tmp/smb$v 12000 2813 | grep 'K1 diag')"
done
