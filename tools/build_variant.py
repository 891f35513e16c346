"""Build the library a second time under another name for A / B runs on one GPU box: ab/liblvba_<name>.so (git-ignored, travels with
the snapshot; LVBA_HIP_LIB=ab/liblvba_<name>.so picks it).  usage: python tools/build_variant.py <name> [extra compiler flags...]"""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("global-lvba_amd.build")


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    objdir = f"/tmp/lvba_variant_{name}"
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.join(ROOT, "ab"), exist_ok=True)
    procs, objs = [], []
    for s in b.SOURCES:
        o = os.path.join(objdir, s.replace(".hip", ".o"))
        procs.append((s, subprocess.Popen([b.hipcc(), *b.flags_for(s), *extra, "-c", os.path.join(b.CSRC, s), "-o", o],
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(o)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise SystemExit(f"hipcc failed on {s}:\n{out}")
    lib = os.path.join(ROOT, "ab", f"liblvba_{name}.so")
    subprocess.check_call([b.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-ldl"])
    print(lib)


if __name__ == "__main__":
    main()
