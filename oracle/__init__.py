"""CPU oracle for the LM-refinement hot path.  TEST INFRASTRUCTURE ONLY (see balm_oracle.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_c(force=False):
    so = os.path.join(_HERE, "libbalm_oracle.so")
    src = os.path.join(_HERE, "balm_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libbalm_oracle.so"])
    return so


def build_voxel(force=False):
    so = os.path.join(_HERE, "libvoxel_oracle.so")
    src = os.path.join(_HERE, "voxel_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libvoxel_oracle.so"])
    return so


_VLIB = None


def voxel_build_cpp(clouds, poses, voxel_size, eigen_ratio=(0.3, 0.1, 0.06, 0.03), min_ps=15):
    """The C++ restatement of cut_voxel / recut / tras_opt (oracle/voxel_oracle.cpp) on a list of [n,>=3] fp32 clouds.
    Returns dict(off, idx, clu, key, n_roots, n_planes, seconds) -- `seconds` is the wall time of the build alone."""
    import time
    global _VLIB
    if _VLIB is None:
        lib = ctypes.CDLL(build_voxel())
        lib.vo_build.restype = ctypes.c_void_p
        lib.vo_build.argtypes = [ctypes.c_int, np.ctypeslib.ndpointer(np.int64, flags="C"),
                                 np.ctypeslib.ndpointer(np.float32, flags="C"), np.ctypeslib.ndpointer(np.float64, flags="C"),
                                 ctypes.c_double, np.ctypeslib.ndpointer(np.float32, flags="C"), ctypes.c_int]
        lib.vo_sizes.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int64)] * 4
        lib.vo_export.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4
        lib.vo_free.argtypes = [ctypes.c_void_p]
        _VLIB = lib
    lib = _VLIB
    off = np.zeros(len(clouds) + 1, np.int64)
    off[1:] = np.cumsum([len(c) for c in clouds])
    pts = np.ascontiguousarray(np.concatenate([np.asarray(c, np.float32)[:, :3] for c in clouds]), np.float32)
    poses = np.ascontiguousarray(poses, np.float64).reshape(-1)
    ratio = np.ascontiguousarray(eigen_ratio, np.float32)
    t0 = time.perf_counter()
    h = lib.vo_build(len(clouds), off, pts.reshape(-1), poses, float(voxel_size), ratio, int(min_ps))
    dt = time.perf_counter() - t0
    n = [ctypes.c_int64() for _ in range(4)]
    lib.vo_sizes(h, *[ctypes.byref(x) for x in n])
    R, NP, V, F = (x.value for x in n)
    o, i, c, k = np.zeros(V + 1, np.int64), np.zeros(F, np.int32), np.zeros((F, 10)), np.zeros((V, 4), np.int64)
    lib.vo_export(h, o.ctypes.data, i.ctypes.data, c.ctypes.data, k.ctypes.data)
    lib.vo_free(h)
    return dict(off=o, idx=i, clu=c, key=k, n_roots=R, n_planes=NP, seconds=dt)


def load_c():
    """ctypes handle to oracle/libbalm_oracle.so (built on demand)."""
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build_c())
        i64p = np.ctypeslib.ndpointer(np.int64, flags="C")
        i32p = np.ctypeslib.ndpointer(np.int32, flags="C")
        f64p = np.ctypeslib.ndpointer(np.float64, flags="C")
        c_i64, c_int, c_dbl = ctypes.c_int64, ctypes.c_int, ctypes.c_double
        dblp, intp, i64ptr = ctypes.POINTER(c_dbl), ctypes.POINTER(c_int), ctypes.POINTER(c_i64)
        prob = [c_int, c_i64, i64p, i32p, f64p, f64p]
        lib.bo_cost.argtypes = prob + [c_int, dblp]
        lib.bo_voxel_lambdas.argtypes = [c_i64, i64p, i32p, f64p, f64p, f64p]
        lib.bo_eval_dense.argtypes = prob + [f64p, f64p, dblp]
        lib.bo_eval_sparse.argtypes = prob + [c_int, c_i64, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_void_p, i64ptr, f64p, dblp]
        lib.bo_ldlt_solve_dense.argtypes = [c_i64, f64p, f64p, f64p, c_int]
        lib.bo_ldlt_solve_band.argtypes = [c_i64, c_i64, f64p, c_i64, f64p, f64p, c_int]
        lib.bo_retract.argtypes = [c_int, f64p, f64p, f64p]
        lib.bo_damping_iter.argtypes = [c_int, c_i64, i64p, i32p, f64p, f64p, c_int, c_dbl, c_dbl,
                                        c_dbl, f64p, intp]
        lib.bo_damping_iter_band.argtypes = [c_int, c_i64, i64p, i32p, f64p, f64p, i32p, c_int, c_int, c_dbl, c_dbl,
                                             c_dbl, c_int, c_int, f64p, intp, f64p]
        _LIB = lib
    return _LIB


class COracle:
    """Thin numpy wrapper over the C restatement for one packed problem."""

    def __init__(self, n_poses, voxel_off, pose_idx, clusters):
        self.lib = load_c()
        self.N = int(n_poses)
        self.voff = np.ascontiguousarray(voxel_off, np.int64)
        self.pidx = np.ascontiguousarray(pose_idx, np.int32)
        self.clu = np.ascontiguousarray(clusters, np.float64).reshape(-1, 10)
        self.V = len(self.voff) - 1

    def _p(self, poses):
        return np.ascontiguousarray(poses, np.float64).reshape(-1, 12)

    def cost(self, poses, avg=False, nthreads=1):
        out = ctypes.c_double()
        self.lib.bo_cost(self.N, self.V, self.voff, self.pidx, self.clu, self._p(poses), nthreads,
                         ctypes.byref(out))
        return out.value / self.V if avg else out.value

    def voxel_lambdas(self, poses):
        lam = np.empty((self.V, 3))
        self.lib.bo_voxel_lambdas(self.V, self.voff, self.pidx, self.clu, self._p(poses), lam)
        return lam

    def eval_dense(self, poses):
        n = 6 * self.N
        H = np.empty((n, n))  # symmetric, so row/col-major agree
        g = np.empty(n)
        c = ctypes.c_double()
        self.lib.bo_eval_dense(self.N, self.V, self.voff, self.pidx, self.clu, self._p(poses), H, g,
                               ctypes.byref(c))
        return H.T, g, c.value

    def dense_to_triplets(self, H, u=0.01):
        """bavoxel.hpp:692-703 on the dense Hessian eval_dense returned (col-major == row-major: symmetric): the dense D and
        HessuD = Hess + u D, then the scan of all (6N)^2 entries into a triplet list.  Returns the number of triplets."""
        n = 6 * self.N
        Hc = np.ascontiguousarray(H)
        self.lib.bo_dense_to_triplets.restype = ctypes.c_int64
        self.lib.bo_dense_to_triplets.argtypes = [ctypes.c_int64, np.ctypeslib.ndpointer(np.float64, flags="C"), ctypes.c_double]
        return int(self.lib.bo_dense_to_triplets(n, Hc, u))

    def eval_sparse(self, poses, nthreads=16, want_blocks=True):
        n = 6 * self.N
        g = np.empty(n)
        c = ctypes.c_double()
        nb = ctypes.c_int64()
        self.lib.bo_eval_sparse(self.N, self.V, self.voff, self.pidx, self.clu, self._p(poses), nthreads,
                                0, None, None, None, ctypes.byref(nb), g, ctypes.byref(c))
        if not want_blocks:
            return None, None, None, g, c.value
        cap = nb.value
        bi = np.empty(cap, np.int32)
        bj = np.empty(cap, np.int32)
        blocks = np.empty((cap, 6, 6))
        rc = self.lib.bo_eval_sparse(self.N, self.V, self.voff, self.pidx, self.clu, self._p(poses),
                                     nthreads, cap, bi.ctypes.data, bj.ctypes.data, blocks.ctypes.data,
                                     ctypes.byref(nb), g, ctypes.byref(c))
        assert rc == 0
        return bi, bj, blocks, g, c.value

    def retract(self, poses, dx):
        out = np.empty((self.N, 12))
        self.lib.bo_retract(self.N, self._p(poses), np.ascontiguousarray(dx, np.float64), out)
        return out

    def damping_iter(self, poses, max_iter=10, u0=0.01, v0=2.0, rel_tol=1e-6):
        x = self._p(poses).copy()
        trace = np.zeros((max_iter, 9))
        nt = ctypes.c_int()
        rc = self.lib.bo_damping_iter(self.N, self.V, self.voff, self.pidx, self.clu, x, max_iter, u0, v0,
                                      rel_tol, trace, ctypes.byref(nt))
        return x, trace[:nt.value], rc


    def damping_iter_band(self, poses, perm=None, max_iter=10, u0=0.01, v0=2.0, rel_tol=1e-6, eval_threads=16,
                          solve_threads=None):
        """damping_iter at sizes where the dense (6N)^2 Hessian is out of reach: sparse block evaluation + unpivoted band
        LDL^T under the pose order `perm` (perm[position] = caller pose; None = natural order).  Returns (poses, trace, rc,
        seconds) with seconds = dict(eval, solve, cost) summed over the iterations."""
        x = self._p(poses).copy()
        perm = np.arange(self.N, dtype=np.int32) if perm is None else np.ascontiguousarray(perm, np.int32)
        iperm = np.empty(self.N, np.int32)
        iperm[perm] = np.arange(self.N, dtype=np.int32)
        # block half-bandwidth of the co-visibility pattern under that order
        pos = iperm[self.pidx].astype(np.int64)
        k = np.diff(self.voff)
        nz = k > 0
        starts = self.voff[:-1][nz]
        bwb = int((np.maximum.reduceat(pos, starts) - np.minimum.reduceat(pos, starts)).max()) if len(pos) else 0
        trace = np.zeros((max_iter, 9))
        nt = ctypes.c_int()
        times = np.zeros(3)
        if solve_threads is None:
            solve_threads = min(32, os.cpu_count() or 1)
        rc = self.lib.bo_damping_iter_band(self.N, self.V, self.voff, self.pidx, self.clu, x, iperm, bwb, max_iter, u0, v0,
                                           rel_tol, int(eval_threads), int(solve_threads), trace, ctypes.byref(nt), times)
        return x, trace[:nt.value], rc, dict(eval=times[0], solve=times[1], cost=times[2], band_blocks=bwb)


def block_parity(H, bi, bj, blocks):
    """Checker for a dense symmetric Hessian H [6N,6N] (from the path under test) against the block list of
    COracle.eval_sparse (6x6 row-major, each unordered pose pair once).  Returns (worst per-block relative error, relative
    weight of everything in H outside the listed blocks)."""
    n = H.shape[0]
    N = n // 6
    Hv = H.reshape(N, 6, N, 6)
    got = Hv[bi, :, bj, :]                                   # [nb, 6, 6]: got[b, r, c] = H[6 bi + r, 6 bj + c]
    scale = np.abs(blocks).reshape(len(bi), -1).max(1)
    err = np.abs(got - blocks).reshape(len(bi), -1).max(1)
    floor = 1e-6 * np.abs(blocks).max()                      # blocks 1e6 x smaller than the largest: absolute bar
    worst = float((err / np.maximum(scale, floor)).max())
    w = np.where(bi == bj, 1.0, 2.0)                         # ||H||_F^2 = sum over listed blocks (off-diagonal ones twice)
    listed = float((w * (got.reshape(len(bi), -1) ** 2).sum(1)).sum())
    total = float((H ** 2).sum())
    return worst, abs(total - listed) / total


def block_parity_sparse(gi, gj, gblocks, bi, bj, blocks, n_poses, details=False):
    """The same check for a path under test that hands out its Hessian as a block list (gi >= gj, gblocks[k, r, c] =
    H[6 gi + r, 6 gj + c], each unordered pair once -- lvba_balm_eval_blocks) against COracle.eval_sparse's (bi <= bj).
    Returns (worst per-block relative error over the oracle's blocks, relative weight of the blocks the path under test has
    and the oracle has not)."""
    N = int(n_poses)
    gkey = gi.astype(np.int64) * N + gj.astype(np.int64)                  # lower: (larger, smaller)
    okey = bj.astype(np.int64) * N + bi.astype(np.int64)                  # upper (bi <= bj) -> the same key
    go, oo = np.argsort(gkey, kind="stable"), np.argsort(okey, kind="stable")
    gk, ok = gkey[go], okey[oo]
    pos = np.searchsorted(gk, ok)
    found = (pos < len(gk)) & (gk[np.minimum(pos, len(gk) - 1)] == ok)
    want = np.transpose(blocks[oo], (0, 2, 1))                            # H[6 bj + r, 6 bi + c]: the lower-oriented block
    got = np.zeros_like(want)
    got[found] = gblocks[go][pos[found]]
    scale = np.abs(want).reshape(len(ok), -1).max(1)
    err = np.abs(got - want).reshape(len(ok), -1).max(1)
    floor = 1e-6 * np.abs(blocks).max()
    worst = float((err / np.maximum(scale, floor)).max())
    w = np.where(gi == gj, 1.0, 2.0)
    tot = w * (gblocks.reshape(len(gi), -1) ** 2).sum(1)
    matched = np.zeros(len(gk), bool)
    matched[pos[found]] = True
    extra = float(tot[go][~matched].sum())
    if details:  # which block is the worst one (tools/worst_block.py)
        k = int(np.argmax(err / np.maximum(scale, floor)))
        return worst, extra / float(tot.sum()), {"pose_i": int(ok[k] // N), "pose_j": int(ok[k] % N), "block_scale": float(scale[k]),
                                                  "block_abs_err": float(err[k]), "largest_block_scale": float(np.abs(blocks).max()),
                                                  "floor": float(floor)}
    return worst, extra / float(tot.sum())


def ldlt_solve_dense(A, b, nthreads=8):
    """Unpivoted LDL^T solve; A symmetric (its C-order memory read col-major is A^T = A)."""
    lib = load_c()
    A = np.array(A, dtype=np.float64, order="C")  # private copy, destroyed
    n = A.shape[0]
    x = np.empty(n)
    rc = lib.bo_ldlt_solve_dense(n, A.reshape(-1), np.ascontiguousarray(b, np.float64), x, nthreads)
    return x, rc


def ldlt_solve_band(AB, bw, b, nthreads=8):
    """AB: [n, ldab] C-order == LAPACK lower band storage AB[(r-c) + c*ldab]; destroyed copy."""
    lib = load_c()
    AB = np.array(AB, dtype=np.float64, order="C")
    n, ldab = AB.shape
    x = np.empty(n)
    rc = lib.bo_ldlt_solve_band(n, bw, AB.reshape(-1), ldab, np.ascontiguousarray(b, np.float64), x, nthreads)
    return x, rc


# ---------------------------------------------------------------------------------------------------------------------
# oracle/_ref: the reference's own BALM headers compiled against the Eigen / PCL stand-ins of oracle/shim (Makefile
# target `ref`, sources read from /root/reference where they lie).  Used by tests/test_ref_pin.py and
# tests/golden/make_golden.py to pin the restatements above; never by the product.
# ---------------------------------------------------------------------------------------------------------------------
REFERENCE_ROOT = os.environ.get("LVBA_REFERENCE_ROOT", "/root/reference")
_RLIB = None


def build_ref(force=False):
    """Builds oracle/_ref/libbalm_ref.so and oracle/_ref/liblvba_system_ref.so when the reference sources are present; returns
    the path of the former, or None when neither the sources nor a prebuilt library exist (e.g. a checkout without
    /root/reference)."""
    so = os.path.join(_HERE, "_ref", "libbalm_ref.so")
    so_sys = os.path.join(_HERE, "_ref", "liblvba_system_ref.so")
    hdr = os.path.join(REFERENCE_ROOT, "include", "BALM", "bavoxel.hpp")
    if os.path.exists(hdr):
        deps = [os.path.join(_HERE, f) for f in ("ref_glue.cpp", "ref_glue_visual.cpp", "ref_glue_system.cpp", "Makefile")]
        for root, _, files in os.walk(os.path.join(_HERE, "shim")):
            deps += [os.path.join(root, f) for f in files]
        deps += [hdr] + [os.path.join(REFERENCE_ROOT, *q) for q in (("include", "BALM", "tools.hpp"), ("include", "utils.hpp"),
                                                                    ("include", "lvba_system.h"), ("include", "dataset_io.h"),
                                                                    ("src", "lvba_system.cpp"), ("src", "dataset_io.cpp"))]
        newest = max(os.path.getmtime(d) for d in deps)
        if force or any(not os.path.exists(f) or os.path.getmtime(f) < newest for f in (so, so_sys)):
            try:
                subprocess.check_call(["make", "-C", _HERE, "-B", "-s", "ref", "REF=" + REFERENCE_ROOT])
            except subprocess.CalledProcessError:
                # the system library (links the host's libsqlite3.so.0) is optional: tests/test_ref_system.py skips without it
                subprocess.check_call(["make", "-C", _HERE, "-B", "-s", "_ref/libbalm_ref.so", "REF=" + REFERENCE_ROOT])
    return so if os.path.exists(so) else None


def ref_system_path():
    """Path of oracle/_ref/liblvba_system_ref.so (built by build_ref), or None."""
    build_ref()
    so = os.path.join(_HERE, "_ref", "liblvba_system_ref.so")
    return so if os.path.exists(so) else None


def build_dropin(force=False):
    """oracle/_ref/liblvba_system_dropin.so: the reference's own src/lvba_system.cpp + src/dataset_io.cpp compiled like
    liblvba_system_ref.so, but with BALM2::damping_iter and ceres::Solve going to the product's liblvba_hip.so through
    include/lvba_adapter.hpp (ref_glue_system.cpp, LVBA_DROPIN).  Built when the reference sources and the product library
    are present; returns the path, or None."""
    so = os.path.join(_HERE, "_ref", "liblvba_system_dropin.so")
    src = os.path.join(REFERENCE_ROOT, "src", "lvba_system.cpp")
    hip = os.path.join(os.path.dirname(_HERE), "global-lvba_amd", "liblvba_hip.so")
    if os.path.exists(src) and os.path.exists(hip):
        deps = [os.path.join(_HERE, "ref_glue_system.cpp"), os.path.join(_HERE, "Makefile"), src,
                os.path.join(os.path.dirname(_HERE), "include", "lvba_adapter.hpp"),
                os.path.join(os.path.dirname(_HERE), "include", "lvba_hip.h")]
        for root, _, files in os.walk(os.path.join(_HERE, "shim")):
            deps += [os.path.join(root, f) for f in files]
        if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
            try:
                subprocess.check_call(["make", "-C", _HERE, "-B", "-s", "dropin", "REF=" + REFERENCE_ROOT])
            except subprocess.CalledProcessError:
                return so if os.path.exists(so) else None
    return so if os.path.exists(so) else None


def load_ref():
    """ctypes handle to oracle/_ref/libbalm_ref.so, or None when it cannot be had."""
    global _RLIB
    if _RLIB is None:
        so = build_ref()
        if so is None:
            return None
        lib = ctypes.CDLL(so)
        f64p = np.ctypeslib.ndpointer(np.float64, flags="C")
        f32p = np.ctypeslib.ndpointer(np.float32, flags="C")
        i64p = np.ctypeslib.ndpointer(np.int64, flags="C")
        c_i64, c_int, c_dbl, vp = ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_void_p
        lib.ref_acc_evaluate2.restype = c_i64
        lib.ref_acc_evaluate2.argtypes = [c_int, c_i64, f64p, f64p, f64p, f64p, ctypes.POINTER(c_dbl)]
        lib.ref_divide_thread.restype = c_dbl
        lib.ref_divide_thread.argtypes = [c_int, c_i64, f64p, f64p, f64p, f64p]
        lib.ref_only_residual.restype = c_dbl
        lib.ref_only_residual.argtypes = [c_int, c_i64, f64p, f64p, c_int]
        lib.ref_damping_iter.restype = c_i64
        lib.ref_damping_iter.argtypes = [c_int, c_i64, f64p, f64p]
        lib.ref_exp.argtypes = [f64p, f64p]
        lib.ref_transform_cluster.argtypes = [f64p, f64p, f64p]
        lib.ref_map_build.restype = vp
        lib.ref_map_build.argtypes = [c_int, i64p, f32p, f64p, c_dbl, f32p]
        lib.ref_map_sizes.argtypes = [vp] + [ctypes.POINTER(c_i64)] * 3
        lib.ref_map_export.argtypes = [vp, i64p, f64p, f64p]
        lib.ref_map_find_planes.argtypes = [vp, c_i64, f64p, c_dbl, f64p]
        lib.ref_map_free.argtypes = [vp]
        lib.ref_down_sampling_voxel2.restype = c_i64
        lib.ref_down_sampling_voxel2.argtypes = [c_i64, f32p, c_dbl, f32p]
        lib.ref_pl_transform.argtypes = [c_i64, f32p, f64p]
        i32p = np.ctypeslib.ndpointer(np.int32, flags="C")
        lib.ref_reproj.argtypes = [f64p, f64p, f64p, f64p, f64p, c_dbl, c_dbl, f64p, f64p]
        u8p = np.ctypeslib.ndpointer(np.uint8, flags="C")
        if hasattr(lib, "ref_visual_jacobian_pass"):
            lib.ref_visual_jacobian_pass.restype = c_dbl
            lib.ref_visual_jacobian_pass.argtypes = [c_i64, i64p, i32p, f64p, f64p, f64p, f64p, f64p, u8p, f64p, c_dbl, c_dbl, c_int,
                                                     ctypes.POINTER(c_dbl), ctypes.POINTER(c_dbl)]
        if hasattr(lib, "ref_visual_reduced_system"):
            lib.ref_visual_reduced_system.restype = c_int
            lib.ref_visual_reduced_system.argtypes = [c_int, c_i64, i64p, i32p, f64p, f64p, f64p, f64p, f64p, u8p, f64p, c_dbl, c_dbl,
                                                      c_dbl, c_dbl, c_dbl, c_int, c_int, f64p, f64p, ctypes.POINTER(c_dbl), f64p]
        lib.ref_plane.argtypes = [f64p, c_dbl, c_dbl, f64p, f64p, f64p]
        for fn in (lib.ref_distort, lib.ref_undistort):
            fn.restype, fn.argtypes = c_int, [f64p, c_dbl, c_dbl, f64p]
        lib.ref_project_world.restype, lib.ref_project_world.argtypes = c_int, [f64p, f64p, f64p, f64p, f64p]
        lib.ref_backproject.restype, lib.ref_backproject.argtypes = c_int, [f64p, c_dbl, c_dbl, c_dbl, f64p]
        lib.ref_cam_to_world.argtypes = [f64p, f64p, f64p, f64p]
        lib.ref_pair_index.restype, lib.ref_pair_index.argtypes = c_i64, [c_int, c_int, c_int]
        lib.ref_compute_mad.restype, lib.ref_compute_mad.argtypes = c_dbl, [c_i64, f64p]
        lib.ref_pick_largest_cluster.restype = c_i64
        lib.ref_pick_largest_cluster.argtypes = [c_i64, f64p, c_i64, i32p, i32p]
        lib.ref_euler_to_rot.argtypes = [c_dbl, c_dbl, c_dbl, f64p]
        lib.ref_parse_timestamp.restype, lib.ref_parse_timestamp.argtypes = c_int, [ctypes.c_char_p, ctypes.POINTER(c_dbl)]
        lib.ref_fetch_depth_bilinear.restype = c_int
        lib.ref_fetch_depth_bilinear.argtypes = [c_int, c_int, f32p, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.POINTER(ctypes.c_float)]
        _RLIB = lib
    return _RLIB


def csr_to_slots(n_poses, voxel_off, pose_idx, clusters):
    """CSR problem (lvba_balm_create layout) -> the reference's dense layout [V][win_size][10] with empty slots zero."""
    voxel_off = np.asarray(voxel_off, np.int64)
    V = len(voxel_off) - 1
    out = np.zeros((V, int(n_poses), 10))
    clusters = np.asarray(clusters, np.float64).reshape(-1, 10)
    for a in range(V):
        for f in range(voxel_off[a], voxel_off[a + 1]):
            out[a, int(pose_idx[f])] = clusters[f]
    return out


class Reference:
    """numpy front of oracle/_ref/libbalm_ref.so (the reference's BALM code).  Reference.available() first."""

    @staticmethod
    def available():
        return load_ref() is not None

    def __init__(self):
        self.lib = load_ref()
        if self.lib is None:
            raise RuntimeError("oracle/_ref/libbalm_ref.so is absent and /root/reference is not there to build it from")

    @staticmethod
    def _c(a, dt=np.float64):
        return np.ascontiguousarray(a, dt)

    def exp(self, w):
        R = np.empty(9)
        self.lib.ref_exp(self._c(w).reshape(3), R)
        return R.reshape(3, 3)

    def transform_cluster(self, cluster, pose):
        out = np.empty(10)
        self.lib.ref_transform_cluster(self._c(cluster).reshape(10), self._c(pose).reshape(12), out)
        return out

    def acc_evaluate2(self, slots, poses):
        V, win = slots.shape[:2]
        H, g, r = np.empty((6 * win, 6 * win)), np.empty(6 * win), ctypes.c_double()
        n = self.lib.ref_acc_evaluate2(win, V, self._c(slots).reshape(-1), self._c(poses).reshape(-1), H, g, ctypes.byref(r))
        return H, g, r.value, int(n)

    def divide_thread(self, slots, poses):
        V, win = slots.shape[:2]
        H, g = np.empty((6 * win, 6 * win)), np.empty(6 * win)
        r = self.lib.ref_divide_thread(win, V, self._c(slots).reshape(-1), self._c(poses).reshape(-1), H, g)
        return H, g, float(r)

    def only_residual(self, slots, poses, is_avg=False):
        V, win = slots.shape[:2]
        return float(self.lib.ref_only_residual(win, V, self._c(slots).reshape(-1), self._c(poses).reshape(-1), int(is_avg)))

    def damping_iter(self, slots, poses):
        V, win = slots.shape[:2]
        x = np.array(poses, np.float64).reshape(-1).copy()
        self.lib.ref_damping_iter(win, V, self._c(slots).reshape(-1), x)
        return x.reshape(-1, 12)

    def visual_jacobian_pass(self, q, t, X, obs_off, obs_cam, obs_uv, plane, valid, intr, sigma_px=0.5, sigma_plane=0.01,
                             nthreads=None):
        """One residual + Jet-Jacobian pass of the reference's own cost functors over a visual problem (ref_glue_visual.cpp).
        Returns (seconds, cost)."""
        if nthreads is None:
            nthreads = os.cpu_count() or 1
        c, js = ctypes.c_double(), ctypes.c_double()
        sec = self.lib.ref_visual_jacobian_pass(len(obs_off) - 1, self._c(obs_off, np.int64), self._c(obs_cam, np.int32),
                                                self._c(obs_uv).reshape(-1), self._c(q).reshape(-1), self._c(t).reshape(-1),
                                                self._c(X).reshape(-1), self._c(plane).reshape(-1), self._c(valid, np.uint8),
                                                self._c(intr), float(sigma_px), float(max(1e-9, sigma_plane)), int(nthreads),
                                                ctypes.byref(c), ctypes.byref(js))
        return float(sec), c.value

    def visual_reduced_system(self, q, t, X, obs_off, obs_cam, obs_uv, plane, valid, intr, radius=1e4, sigma_px=0.5,
                              sigma_plane=0.01, kb=3, min_diag=1e-6, max_diag=1e32, nthreads=None):
        """The reduced camera system of one LM step from the reference's own functors + Jets (ref_glue_visual.cpp:
        ref_visual_reduced_system; the Ceres-internal steps between functors and solver are restated there).  Returns
        (Sband [M, kb+1, 6, 6] lower blocks S[a, a-d], rhs [6M], cost, widest camera pair, Jacobi scales of the camera columns)."""
        if nthreads is None:
            nthreads = min(64, os.cpu_count() or 1)
        M = len(q)
        Sb = np.zeros((M, kb + 1, 6, 6))
        rhs = np.zeros(6 * M)
        scale = np.zeros(6 * M)
        c = ctypes.c_double()
        far = self.lib.ref_visual_reduced_system(M, len(obs_off) - 1, self._c(obs_off, np.int64), self._c(obs_cam, np.int32),
                                                 self._c(obs_uv).reshape(-1), self._c(q).reshape(-1), self._c(t).reshape(-1),
                                                 self._c(X).reshape(-1), self._c(plane).reshape(-1), self._c(valid, np.uint8),
                                                 self._c(intr), float(sigma_px), float(max(1e-9, sigma_plane)), float(radius),
                                                 float(min_diag), float(max_diag), int(kb), int(nthreads), Sb.reshape(-1), rhs,
                                                 ctypes.byref(c), scale)
        return Sb, rhs, c.value, int(far), scale

    def map_build(self, clouds, poses, voxel_size, eigen_ratio=(0.3, 0.1, 0.06, 0.03)):
        """cut_voxel + recut + tras_opt as the reference's call sites run them.  Returns dict(keys [P,4] (x, y, z,
        layer<<6|o1<<3|o2), clusters [P,win,10], geo [P,9] (center, direct, eigenvalues), n_roots, n_admitted, handle),
        planes sorted by key; close with map_free(handle)."""
        win = len(clouds)
        off = np.zeros(win + 1, np.int64)
        off[1:] = np.cumsum([len(c) for c in clouds])
        pts = np.ascontiguousarray(np.concatenate([np.asarray(c, np.float32)[:, :3] for c in clouds]), np.float32)
        import time
        poses_c, ratio_c = self._c(poses).reshape(-1), self._c(eigen_ratio, np.float32)
        t0 = time.perf_counter()
        h = self.lib.ref_map_build(win, off, pts.reshape(-1), poses_c, float(voxel_size), ratio_c)
        seconds = time.perf_counter() - t0                                   # cut_voxel + recut + tras_opt (+ the export walk)
        n = [ctypes.c_int64() for _ in range(3)]
        self.lib.ref_map_sizes(h, *[ctypes.byref(x) for x in n])
        R, P, A = (x.value for x in n)
        keys, clu, geo = np.zeros((P, 4), np.int64), np.zeros((P, win, 10)), np.zeros((P, 9))
        self.lib.ref_map_export(h, keys.reshape(-1), clu.reshape(-1), geo.reshape(-1))
        order = np.lexsort((keys[:, 3], keys[:, 2], keys[:, 1], keys[:, 0]))
        return dict(keys=keys[order], clusters=clu[order], geo=geo[order], n_roots=R, n_admitted=A, handle=h, seconds=seconds)

    def map_find_planes(self, handle, X, voxel_size):
        X = self._c(X).reshape(-1, 3)
        out = np.zeros((len(X), 4))
        self.lib.ref_map_find_planes(handle, len(X), X.reshape(-1), float(voxel_size), out.reshape(-1))
        return out

    def map_free(self, handle):
        self.lib.ref_map_free(handle)

    def down_sampling_voxel2(self, pts, leaf):
        pts = self._c(pts, np.float32).reshape(-1, 3)
        out = np.empty_like(pts)
        n = self.lib.ref_down_sampling_voxel2(len(pts), pts.reshape(-1), float(leaf), out.reshape(-1))
        return out[:n]

    def pl_transform(self, pts, pose):
        p = np.array(pts, np.float32).reshape(-1, 3).copy()
        self.lib.ref_pl_transform(len(p), p.reshape(-1), self._c(pose).reshape(12))
        return p

    # ---- include/utils.hpp (oracle/ref_glue_visual.cpp)
    def reproj(self, q, t, X, uv, intr, su=1.0, sv=1.0):
        """ReprojErrorWhitenedDistorted: (r [2], J [2,10] = d r / d (q[4] as stored [w,x,y,z], t[3], X[3]))."""
        r, J = np.empty(2), np.empty(20)
        self.lib.ref_reproj(self._c(q).reshape(4), self._c(t).reshape(3), self._c(X).reshape(3), self._c(uv).reshape(2),
                            self._c(intr).reshape(8), float(su), float(sv), r, J)
        return r, J.reshape(2, 10)

    def plane(self, n, d, sigma, X):
        r, J = np.empty(1), np.empty(3)
        self.lib.ref_plane(self._c(n).reshape(3), float(d), float(sigma), self._c(X).reshape(3), r, J)
        return r[0], J

    def distort(self, intr, x, y):
        out = np.zeros(2)
        return bool(self.lib.ref_distort(self._c(intr).reshape(8), float(x), float(y), out)), out

    def undistort(self, intr, u, v):
        out = np.zeros(2)
        return bool(self.lib.ref_undistort(self._c(intr).reshape(8), float(u), float(v), out)), out

    def project_world(self, intr, Rcw, tcw, Xw):
        out = np.zeros(3)
        ok = self.lib.ref_project_world(self._c(intr).reshape(8), self._c(Rcw).reshape(9), self._c(tcw).reshape(3),
                                        self._c(Xw).reshape(3), out)
        return bool(ok), out

    def backproject(self, intr, u, v, depth):
        out = np.zeros(3)
        return bool(self.lib.ref_backproject(self._c(intr).reshape(8), float(u), float(v), float(depth), out)), out

    def cam_to_world(self, Xc, Rcw, tcw):
        out = np.zeros(3)
        self.lib.ref_cam_to_world(self._c(Xc).reshape(3), self._c(Rcw).reshape(9), self._c(tcw).reshape(3), out)
        return out

    def pair_index(self, i, j, N):
        return int(self.lib.ref_pair_index(int(i), int(j), int(N)))

    def compute_mad(self, resid):
        resid = self._c(resid).reshape(-1)
        return float(self.lib.ref_compute_mad(len(resid), resid))

    def pick_largest_cluster(self, pts, idx_valid):
        pts = self._c(pts).reshape(-1, 3)
        iv = np.ascontiguousarray(idx_valid, np.int32)
        out = np.zeros(max(1, len(iv)), np.int32)
        n = self.lib.ref_pick_largest_cluster(len(pts), pts.reshape(-1), len(iv), iv, out)
        return out[:n].copy()

    def euler_to_rot(self, roll, pitch, yaw):
        R = np.empty(9)
        self.lib.ref_euler_to_rot(float(roll), float(pitch), float(yaw), R)
        return R.reshape(3, 3)

    def parse_timestamp(self, name):
        ts = ctypes.c_double()
        ok = self.lib.ref_parse_timestamp(name.encode(), ctypes.byref(ts))
        return (ts.value if ok else None)

    def fetch_depth_bilinear(self, depth, u, v, depth_scale=0.001):
        depth = self._c(depth, np.float32)
        out = ctypes.c_float()
        ok = self.lib.ref_fetch_depth_bilinear(depth.shape[0], depth.shape[1], depth.reshape(-1), float(u), float(v),
                                               float(depth_scale), ctypes.byref(out))
        return out.value if ok else None
