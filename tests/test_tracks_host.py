"""CPU tests (no GPU) of the per-track device code: global-lvba_amd/csrc/tracks_device.h (DLT, mean reprojection, camera
model) and fusion_device.h (depth sampling, depth-fused and triangulated candidates, selection) are host/device-neutral, so
the very functions the kernels call are compiled with g++ (tests/host_emul_tracks.cpp) and held against
oracle/track_oracle.py / oracle/fusion_oracle.py here -- the same comparisons tests/test_gpu_fusion.py and
tests/test_gpu_visual.py make on the GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import fusion_oracle as fo
from oracle import track_oracle as to

import test_gpu_fusion as G   # scene / track generators (pure numpy; its tests are GPU-marked, the helpers are not)


def build_emul(directory):
    """Compiles tests/host_emul_tracks.cpp (the device headers, for the host) into `directory` and binds it."""
    so = os.path.join(str(directory), "libemul_tracks.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                           os.path.join(ROOT, "tests", "host_emul_tracks.cpp"), "-o", so])
    lib = ctypes.CDLL(so)
    f64 = np.ctypeslib.ndpointer(np.float64, flags="C")
    f32 = np.ctypeslib.ndpointer(np.float32, flags="C")
    i64 = np.ctypeslib.ndpointer(np.int64, flags="C")
    i32 = np.ctypeslib.ndpointer(np.int32, flags="C")
    u8 = np.ctypeslib.ndpointer(np.uint8, flags="C")
    lib.emul_triangulate.argtypes = [ctypes.c_int64, i64, i32, f64, f64, f64, ctypes.c_int32, f64, f64, f64, i32, u8]
    lib.emul_fuse_tracks.argtypes = [ctypes.c_int64, i64, i32, f32, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, f64, f64,
                                     ctypes.c_int32, f64, ctypes.c_int, ctypes.c_double, ctypes.c_double, u8, f64, f64, u8]
    lib.emul_umap_order.restype = ctypes.c_int
    lib.emul_umap_order.argtypes = [ctypes.c_int, ctypes.c_int, i32, i32]
    lib.emul_fetch_depth.restype = ctypes.c_float
    lib.emul_fetch_depth.argtypes = [f32, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.POINTER(ctypes.c_int)]
    return lib


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    return build_emul(tmp_path_factory.mktemp("emul_tracks"))


def _fuse(lib, off, img, uv, depth, Rcw, tcw, intr, obser_thr=3, angle=8.0, thr=3.0):
    n, O = len(off) - 1, len(img)
    st, X, err, kept = np.zeros(n, np.uint8), np.zeros((n, 3)), np.zeros(n), np.zeros(max(O, 1), np.uint8)
    d = None if depth is None else np.ascontiguousarray(depth, np.float32)
    lib.emul_fuse_tracks(n, np.ascontiguousarray(off, np.int64), np.ascontiguousarray(img, np.int32),
                         np.ascontiguousarray(uv, np.float32).reshape(-1), None if d is None else d.ctypes.data,
                         0 if d is None else d.shape[2], 0 if d is None else d.shape[1],
                         np.ascontiguousarray(Rcw, np.float64).reshape(-1), np.ascontiguousarray(tcw, np.float64).reshape(-1),
                         len(Rcw), np.ascontiguousarray(intr, np.float64), obser_thr, float(np.cos(np.radians(angle))), thr,
                         st, X.reshape(-1), err, kept)
    return st, X, err, kept[:O]


def test_device_fusion_code_matches_oracle(emul):
    clouds, poses, times, img_t, Rcw, tcw = G._scene(n_frames=6, pts=20000)
    rng = np.random.default_rng(8)
    off, img, uv, X = G._tracks(clouds, poses, Rcw, tcw, rng, n_tracks=240)
    depth = fo.render_depth(clouds, poses, times, img_t, Rcw, tcw, G.INTR, G.W, G.H, half_w=100.0)
    for dimg in (depth, None):
        want = fo.fuse_tracks(off, img, uv, dimg, Rcw, tcw, G.INTR)
        got = _fuse(emul, off, img, uv, dimg, Rcw, tcw, G.INTR)
        np.testing.assert_array_equal(got[0], want[0])
        np.testing.assert_array_equal(got[3], want[3])
        ok = want[0] > 0
        assert np.abs(got[1][ok] - want[1][ok]).max() <= 1e-9 and np.abs(got[2][ok] - want[2][ok]).max() <= 1e-9
        assert np.isinf(got[2][~ok]).all() and not got[1][~ok].any()
    st = fo.fuse_tracks(off, img, uv, depth, Rcw, tcw, G.INTR)[0]
    assert (st == 1).sum() >= 15 and (st == 2).sum() >= 10 and (st == 0).sum() >= 1
    # other thresholds: a looser angle / stricter reprojection gate move tracks between the outcomes identically on both sides
    for kw in (dict(angle=2.0), dict(thr=0.4), dict(obser_thr=4)):
        want = fo.fuse_tracks(off, img, uv, depth, Rcw, tcw, G.INTR, obser_thr=kw.get("obser_thr", 3),
                              min_view_angle_deg=kw.get("angle", 8.0), reproj_thr=kw.get("thr", 3.0))
        got = _fuse(emul, off, img, uv, depth, Rcw, tcw, G.INTR, **kw)
        np.testing.assert_array_equal(got[0], want[0])


def test_device_dlt_and_depth_sampling_match_oracle(emul):
    clouds, poses, times, img_t, Rcw, tcw = G._scene(n_frames=4, pts=8000, n_cams=6)
    rng = np.random.default_rng(3)
    off, img, uv, X = G._tracks(clouds, poses, Rcw, tcw, rng, n_tracks=80)
    # de-duplicate per image (the stand-alone DLT takes one observation per image)
    o2, i2, u2 = [0], [], []
    for t in range(len(off) - 1):
        seen = set()
        for o in range(off[t], off[t + 1]):
            if int(img[o]) not in seen:
                seen.add(int(img[o])); i2.append(img[o]); u2.append(uv[o].astype(np.float64))
        o2.append(len(i2))
    o2, i2, u2 = np.array(o2, np.int64), np.array(i2, np.int32), np.array(u2, np.float64).reshape(-1, 2)
    n = len(o2) - 1
    Xg, eg, cg, okg = np.zeros((n, 3)), np.zeros(n), np.zeros(n, np.int32), np.zeros(n, np.uint8)
    emul.emul_triangulate(n, o2, i2, u2.reshape(-1), np.ascontiguousarray(Rcw).reshape(-1), np.ascontiguousarray(tcw).reshape(-1),
                          len(Rcw), np.ascontiguousarray(G.INTR, np.float64), Xg.reshape(-1), eg, cg, okg)
    ok, Xo, eo, co = to.triangulate_tracks(G.INTR, Rcw, tcw, o2, i2, u2)
    np.testing.assert_array_equal(okg, ok)
    np.testing.assert_array_equal(cg[ok > 0], co[ok > 0])
    assert np.abs(Xg[ok > 0] - Xo[ok > 0]).max() <= 1e-8 and np.abs(eg[ok > 0] - eo[ok > 0]).max() <= 1e-8
    assert ok.sum() >= 40
    # bilinear depth sampling, float arithmetic without contraction: bit for bit
    depth = (2.0 + rng.random((30, 40))).astype(np.float32)
    depth[7, 9] = 0.0
    for _ in range(400):
        u, v = np.float32(rng.uniform(-1, 41)), np.float32(rng.uniform(-1, 31))
        okc = ctypes.c_int()
        d = emul.emul_fetch_depth(depth.reshape(-1), 40, 30, float(u), float(v), ctypes.byref(okc))
        want = fo.fetch_depth_bilinear(depth, u, v)
        assert bool(okc.value) == (want is not None)
        if want is not None:
            assert np.float32(d) == np.float32(want)


def test_device_umap_order_is_the_real_container_order(emul, tmp_path):
    """umap_order / umap_bucket_count of tracks_device.h (what the kernel walks) against std::unordered_map<int,int> itself."""
    src = tmp_path / "umap.cpp"
    src.write_text('#include <cstddef>\n#include <unordered_map>\nextern "C" int real_order(int res, int m, const int *k, int *out) {\n'
                   "  std::unordered_map<int, int> u; u.reserve((std::size_t)res);\n  for (int i = 0; i < m; ++i) u[k[i]] = i;\n"
                   "  int n = 0; for (const auto &kv : u) out[n++] = kv.second; return (int)u.bucket_count(); }\n")
    so = str(tmp_path / "umap.so")
    subprocess.check_call(["g++", "-O1", "-shared", "-fPIC", "-o", so, str(src)])
    real = ctypes.CDLL(so)
    ip = ctypes.POINTER(ctypes.c_int)
    rng = np.random.default_rng(11)
    for _ in range(800):
        m = int(rng.integers(1, 60))
        res = m + int(rng.integers(0, 50))
        keys = rng.choice(int(rng.integers(m, 5000)), m, replace=False).astype(np.int32)
        want, got = np.zeros(m, np.int32), np.zeros(m, np.int32)
        B = real.real_order(res, m, keys.ctypes.data_as(ip), want.ctypes.data_as(ip))
        assert emul.emul_umap_order(res, m, keys, got) == B
        np.testing.assert_array_equal(got, want)
        assert fo.umap_order(keys.tolist(), res) == want.tolist()
