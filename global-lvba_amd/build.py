"""Build liblvba_hip.so (gfx950) in-tree with hipcc.  No torch, no JIT cache: the .so sits next to
this file so it travels with the repo snapshot to the GPU box."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblvba_hip.so")
SOURCES = ["lvba_api.hip", "block_system.hip", "balm_kernels.hip", "ldlt.hip", "visual_api.hip", "visual_kernels.hip",
           "voxelize.hip", "window_ba.hip", "tracks.hip", "pair_lists.hip", "fusion.hip", "bcr.hip"]
HEADERS = ["balm_math.h", "lvba_internal.h", "lvba_common.h", "block_system.h", "visual_math.h", "mempool.h", "voxel_internal.h", "pair_lists.h", "tracks_device.h", "ordering.h", "host_tables.h", "host_arena.h", "fusion_device.h", "ldlt_lookahead.h", "ldlt_schedule.h", "ldlt_prepare.h", "ldlt_diag.h", "ldlt_tiles.h", "ldlt_back.h", "ldlt_nd.h", "nd_plan.h", "key_pack.h", os.path.join("..", "..", "include", "lvba_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-result", "-Wno-unused-value"]
# Kernels that take DISCRETE decisions on floating-point values (voxel keys, pixel indices, depth / angle / reprojection
# thresholds, fp32 write-backs: the depth renderer, the track fusion, the triangulation, the anchor merge and down-sampling)
# must round like the reference's plain x86-64 build, expression by expression: no contraction of a*b+c into FMAs there.
# The LM kernels (smooth arithmetic, compared at 1e-8) keep the FMAs.
NO_CONTRACT = {"fusion.hip", "tracks.hip", "window_ba.hip"}


def flags_for(src):
    return FLAGS + ["-ffp-contract=off" if src in NO_CONTRACT else "-ffp-contract=fast"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace(".hip", ".o"))
        cmd = [hipcc(), *flags_for(s), "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(o)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
