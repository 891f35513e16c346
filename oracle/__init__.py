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
