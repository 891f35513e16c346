"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol include/lvba_hip.h
declares, and refuses to compute without a HIP device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, make_problem


@pytest.fixture(scope="module")
def lib(pkg):
    import __graft_entry__ as ge
    ge.build()
    return pkg._lib.load()


def test_header_symbols_are_exported(lib, pkg):
    hdr = open(os.path.join(ROOT, "include", "lvba_hip.h")).read()
    declared = set(re.findall(r"\b(lvba_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(pkg._lib.SYMBOLS), declared ^ set(pkg._lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.lvba_version() >= 100


def test_struct_layouts_match_header(pkg):
    L = pkg._lib
    assert ctypes.sizeof(L.BalmOpts) == 32 and ctypes.sizeof(L.LmTrace) == 64
    assert ctypes.sizeof(L.BalmInfo) == 136 and ctypes.sizeof(L.Prof) == 80 and ctypes.sizeof(L.NdModel) == 40
    o = pkg.BalmProblem.default_opts()
    assert (o.max_iter, o.u0, o.v0, o.rel_tol) == (10, 0.01, 2.0, 1e-6)      # bavoxel.hpp:664,686,760


def test_every_struct_has_the_size_the_c_compiler_gives_it(pkg, tmp_path):
    """ctypes mirrors vs `sizeof` from the header itself (compiled as C: the header must stay plain C)."""
    import subprocess
    L = pkg._lib
    pairs = [("lvba_balm_opts", L.BalmOpts), ("lvba_lm_trace", L.LmTrace), ("lvba_balm_info_t", L.BalmInfo),
             ("lvba_prof_t", L.Prof), ("lvba_visual_opts", L.VisualOpts), ("lvba_visual_trace", L.VisualTrace),
             ("lvba_voxel_opts", L.VoxelOpts), ("lvba_voxmap_info_t", L.VoxmapInfo), ("lvba_window_opts", L.WindowOpts),
             ("lvba_window_info", L.WindowInfo), ("lvba_lidar_ba_opts", L.LidarBaOpts), ("lvba_lidar_ba_report", L.LidarBaReport),
             ("lvba_fuse_opts", L.FuseOpts), ("lvba_nd_model_t", L.NdModel)]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "lvba_hip.h"\nint main(void){' +
                   "".join(f'printf("%zu\\n", sizeof({n}));' for n, _ in pairs) + "return 0;}\n")
    exe = str(tmp_path / "sizes")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe])
    sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    for (name, cls), sz in zip(pairs, sizes):
        assert ctypes.sizeof(cls) == sz, (name, ctypes.sizeof(cls), sz)


def test_shard_range_matches_reference_slicing(pkg):
    from oracle import balm_oracle as bo
    for V in (7, 16, 100, 12345, 2_000_000):
        ref = bo.thread_slices(V) if V >= 16 else None
        got = [pkg.shard_range(V, r, 16) for r in range(16)]
        assert got[0][0] == 0 and got[-1][1] == V and all(a[1] == b[0] for a, b in zip(got, got[1:]))
        if ref:
            assert got == ref                                               # bavoxel.hpp:621-624


def test_no_cpu_fallback(lib, pkg):
    if lib.lvba_device_count() > 0:
        pytest.skip("a HIP device is present")
    d = make_problem(12, 60, band=4, seed=1)
    with pytest.raises(pkg._lib.LvbaError) as e:
        pkg.BalmProblem(d["n_poses"], d["voxel_off"], d["pose_idx"], d["clusters"])
    assert e.value.code == pkg._lib.ERR_DEVICE and "no CPU fallback" in str(e.value)


def test_product_does_not_import_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pk = os.path.join(ROOT, "global-lvba_amd")
    for dirpath, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                for pat in (r"^\s*(import|from)\s+oracle", r"libbalm_oracle", r"balm_oracle", r"oracle/", r"import_module\([\"']oracle"):
                    assert not re.search(pat, src, re.M), (os.path.join(dirpath, f), pat)


def test_vox_hess_mirror_packs_like_the_reference(pkg):
    N = 5
    vh = pkg.VOX_HESS(N)
    sig = np.zeros((N, 10)); sig[1, 9] = 20; sig[3, 9] = 15; sig[1, :9] = 1.0; sig[3, :9] = 2.0
    vh.push_voxel(sig)
    one = np.zeros((N, 10)); one[2, 9] = 30
    vh.push_voxel(one)                                   # only one observer: dropped (bavoxel.hpp:52)
    off, idx, clu = vh.pack()
    assert off.tolist() == [0, 2] and idx.tolist() == [1, 3] and clu.shape == (2, 10) and clu[1, 9] == 15


def test_cpp_adapter_compiles_and_links(lib, tmp_path):
    """include/lvba_adapter.hpp (the reference-side binding of INTEGRATION.md) against stand-in types."""
    import subprocess
    exe = str(tmp_path / "adapter_check")
    libdir = os.path.join(ROOT, "global-lvba_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "adapter_check.cpp"), "-o", exe,
                           "-L", libdir, "-llvba_hip", f"-Wl,-rpath,{libdir}"])
    assert subprocess.call([exe]) == 0


def test_window_split_deals_out_whole_windows():
    """lvba_window_split (host only, no device needed): contiguous runs of whole windows, every frame in exactly one share,
    the thread split of bavoxel.hpp:621-624 applied to windows; shares may be empty when there are fewer windows than shares."""
    import ctypes as C
    import importlib
    L = importlib.import_module("global-lvba_amd._lib")
    lib = L.load()
    for n, w, k in ((320, 20, 8), (26, 4, 3), (26, 4, 2), (5, 10, 4), (0, 10, 2), (64, 16, 1), (1000, 7, 8)):
        fb = np.zeros(k + 1, np.int32)
        assert lib.lvba_window_split(n, w, k, fb) == 0
        assert fb[0] == 0 and fb[-1] == n and (np.diff(fb) >= 0).all()
        assert all(int(b) % w == 0 for b in fb[:-1])
        nw = (n + w - 1) // w
        per = [(-(-int(fb[i + 1] - fb[i]) // w)) for i in range(k)]
        assert sum(per) == nw and max(per) - min(per) <= 1
    assert lib.lvba_window_split(10, 0, 2, np.zeros(3, np.int32)) != 0
