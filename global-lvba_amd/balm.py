"""Host-side mirror of the reference's BALM interface over the C-ABI.

`BalmProblem` is the thin object wrapper of an `lvba_balm_t` handle.  `VOX_HESS` / `BALM2` mirror the
reference classes of the same names (include/BALM/bavoxel.hpp:32-269, 587-767) -- same method names,
argument meaning and return values -- so the parity tests read like calls into the reference:

    voxhess = VOX_HESS(win_size); voxhess.push_voxel(sig_orig)            # bavoxel.hpp:45-54
    opt = BALM2(win_size); opt.damping_iter(x_stats, voxhess)              # bavoxel.hpp:662

All arithmetic runs in liblvba_hip.so on the GPU; this file only packs arrays and forwards calls.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def shard_range(n_voxels, rank, n_ranks):
    """[floor(V*r/G), floor(V*(r+1)/G)) -- bavoxel.hpp:621-624 with thread -> GPU."""
    a, b = C.c_int64(), C.c_int64()
    L.load().lvba_shard_range(int(n_voxels), int(rank), int(n_ranks), C.byref(a), C.byref(b))
    return a.value, b.value


class BalmProblem:
    """One packed LiDAR-BA problem (or one rank's voxel shard) resident on a GPU."""

    def __init__(self, n_poses, voxel_off, pose_idx, clusters, device=0, ordering=None, band_frac=None):
        self.lib = L.load()
        voxel_off = np.ascontiguousarray(voxel_off, np.int64)
        pose_idx = np.ascontiguousarray(pose_idx, np.int32)
        clusters = np.ascontiguousarray(clusters, np.float64).reshape(-1, 10)
        self.n_poses = int(n_poses)
        self.n_voxels = len(voxel_off) - 1
        self._h = C.c_void_p()
        # pose_idx / clusters are indexed relative to voxel_off[0]
        L.check(self.lib.lvba_balm_create(self.n_poses, self.n_voxels, voxel_off, pose_idx, clusters.reshape(-1),
                                          int(device), C.byref(self._h)))
        if ordering is not None or band_frac is not None:
            L.check(self.lib.lvba_balm_configure(self._h, 1 if ordering is None else int(ordering),
                                                 0.6 if band_frac is None else float(band_frac)))

    @classmethod
    def _from_handle(cls, handle, n_poses, n_voxels, ordering=None, band_frac=None):
        """Wrap an lvba_balm_t made by another entry point (lvba_voxmap_to_balm)."""
        self = cls.__new__(cls)
        self.lib = L.load()
        self.n_poses, self.n_voxels, self._h = int(n_poses), int(n_voxels), handle
        if ordering is not None or band_frac is not None:
            L.check(self.lib.lvba_balm_configure(self._h, 1 if ordering is None else int(ordering),
                                                 0.6 if band_frac is None else float(band_frac)))
        return self

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.lvba_balm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _poses(self, poses):
        x = np.ascontiguousarray(poses, np.float64).reshape(-1)
        if x.size != 12 * self.n_poses:
            raise ValueError(f"poses must hold {self.n_poses} x 12 doubles")
        return x

    # -- multi-GPU ----------------------------------------------------------------------------------
    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        L.check(L.load().lvba_dist_unique_id(buf))
        return buf.raw

    def dist_init(self, n_ranks, rank, uid):
        L.check(self.lib.lvba_balm_dist_init(self._h, int(n_ranks), int(rank), bytes(uid)))

    def dist_init_external(self, n_ranks, rank, fn, ctx):
        """The caller's all-reduce instead of RCCL (lvba_balm_dist_init_external): fn = address of an lvba_allreduce_fn,
        ctx = its context pointer."""
        L.check(self.lib.lvba_balm_dist_init_external(self._h, int(n_ranks), int(rank), C.c_void_p(fn), C.c_void_p(ctx)))

    # -- queries ------------------------------------------------------------------------------------
    def info(self):
        i = L.BalmInfo()
        L.check(self.lib.lvba_balm_info(self._h, C.byref(i)))
        return {f: getattr(i, f) for f, _ in i._fields_}

    def nd_model(self, n_ranks):
        """the nested-dissection plan of this problem's graph on n_ranks ranks and its cost model (lvba_balm_nd_model)"""
        m = L.NdModel()
        L.check(self.lib.lvba_balm_nd_model(self._h, int(n_ranks), C.byref(m)))
        return {f: getattr(m, f) for f, _ in m._fields_}

    def ordering(self):
        perm = np.empty(self.n_poses, np.int32)
        L.check(self.lib.lvba_balm_get_ordering(self._h, perm))
        return perm

    def set_profiling(self, on=True):
        L.check(self.lib.lvba_balm_set_profiling(self._h, 1 if on else 0))

    def profile(self, reset=False):
        p = L.Prof()
        L.check(self.lib.lvba_balm_get_profile(self._h, C.byref(p), 1 if reset else 0))
        return {f: getattr(p, f) for f, _ in p._fields_}

    # -- the hot path ---------------------------------------------------------------------------------
    def cost(self, poses, is_avg=False):
        out = C.c_double()
        L.check(self.lib.lvba_balm_cost(self._h, self._poses(poses), 1 if is_avg else 0, C.byref(out)))
        return out.value

    def eval(self, poses, want_H=True, want_g=True):
        n = 6 * self.n_poses
        H = np.empty((n, n)) if want_H else None
        g = np.empty(n) if want_g else None
        c = C.c_double()
        L.check(self.lib.lvba_balm_eval(self._h, self._poses(poses), H.ctypes.data if want_H else None,
                                        g.ctypes.data if want_g else None, C.byref(c)))
        return H, g, c.value

    def eval_blocks(self, poses):
        """H in sparse form (lvba_balm_eval_blocks): (bi, bj, blocks [nb, 6, 6], g, cost) with bi >= bj in the caller's pose
        order, every unordered pair once, blocks[k, r, c] = H[6 bi + r, 6 bj + c]."""
        x = self._poses(poses)
        nb = C.c_int64()
        c = C.c_double()
        L.check(self.lib.lvba_balm_eval_blocks(self._h, x, 0, None, None, None, C.byref(nb), None, None))
        cap = int(nb.value)
        bi, bj = np.empty(cap, np.int32), np.empty(cap, np.int32)
        blocks = np.empty((cap, 6, 6))
        g = np.empty(6 * self.n_poses)
        L.check(self.lib.lvba_balm_eval_blocks(self._h, x, cap, bi.ctypes.data, bj.ctypes.data, blocks.ctypes.data, C.byref(nb),
                                               g.ctypes.data, C.byref(c)))
        return bi, bj, blocks, g, c.value

    def solve(self, u):
        dx = np.empty(6 * self.n_poses)
        L.check(self.lib.lvba_balm_solve(self._h, float(u), dx))
        return dx

    @staticmethod
    def default_opts(**kw):
        o = L.BalmOpts()
        L.load().lvba_balm_default_opts(C.byref(o))
        for k, v in kw.items():
            setattr(o, k, v)
        return o

    def refine(self, poses, **opts):
        o = self.default_opts(**opts)
        x = self._poses(poses).copy()
        trace = (L.LmTrace * max(1, o.max_iter))()
        nt = C.c_int32()
        rc = L.check(self.lib.lvba_balm_refine(self._h, x, C.byref(o), trace, C.byref(nt)), allow_numeric=True)
        return x.reshape(-1, 12), [trace[i].as_dict() for i in range(nt.value)], rc

    def set_groups(self, pose_off, voxel_off):
        """Independent groups of poses / voxels (lvba_balm_set_groups); before the first cost / eval / refine call."""
        po = np.ascontiguousarray(pose_off, dtype=np.int32)
        vo = np.ascontiguousarray(voxel_off, dtype=np.int64)
        if len(po) != len(vo) or len(po) < 2:
            raise ValueError("pose_off and voxel_off need n_groups + 1 entries each")
        L.check(self.lib.lvba_balm_set_groups(self._h, len(po) - 1, po, vo))
        self.n_groups = len(po) - 1

    def refine_groups(self, poses, **opts):
        """All groups through one LM loop in lock-step; returns (poses, per-group dict of arrays, rc)."""
        o = self.default_opts(**opts)
        x = self._poses(poses).copy()
        G = self.n_groups
        n_iter, status = np.zeros(G, np.int32), np.zeros(G, np.int32)
        first, last = np.zeros(G), np.zeros(G)
        rc = L.check(self.lib.lvba_balm_refine_groups(self._h, x, C.byref(o), n_iter, status, first, last), allow_numeric=True)
        return x.reshape(-1, 12), dict(n_iter=n_iter, status=status, cost_first=first, cost_last=last), rc

    def lm_begin(self, poses, **opts):
        o = self.default_opts(**opts)
        L.check(self.lib.lvba_balm_lm_begin(self._h, self._poses(poses), C.byref(o)))

    def lm_step(self):
        row, done = L.LmTrace(), C.c_int32()
        rc = L.check(self.lib.lvba_balm_lm_step(self._h, C.byref(row), C.byref(done)), allow_numeric=True)
        return row.as_dict(), bool(done.value), rc

    def lm_end(self, want_poses=True):
        out = np.empty((self.n_poses, 12)) if want_poses else None
        L.check(self.lib.lvba_balm_lm_end(self._h, out.ctypes.data if want_poses else None))
        return out


# =====================================================================================================
# Mirror of the reference classes (same names / call shapes as include/BALM/bavoxel.hpp)
# =====================================================================================================
class IMUST:
    """Pose state; only R, p (and t) are live on this path (tools.hpp:147-207)."""

    def __init__(self, R=None, p=None, t=0.0):
        self.R = np.eye(3) if R is None else np.array(R, dtype=np.float64).reshape(3, 3)
        self.p = np.zeros(3) if p is None else np.array(p, dtype=np.float64).reshape(3)
        self.t = t


def pack_x_stats(x_stats):
    return np.stack([np.concatenate([x.R.reshape(9), x.p]) for x in x_stats])


def unpack_x_stats(poses, x_stats):
    for x, row in zip(x_stats, np.asarray(poses).reshape(-1, 12)):
        x.R = row[:9].reshape(3, 3).copy()
        x.p = row[9:12].copy()


class VOX_HESS:
    """bavoxel.hpp:32-54.  `sig_orig` is one voxel's win_size PointCluster slots as an array
    [win_size, 10] (Pxx Pxy Pxz Pyy Pyz Pzz vx vy vz N); empty slots have N == 0."""

    def __init__(self, win_size=10):
        self.win_size = int(win_size)
        self.plvec_voxels = []

    def push_voxel(self, sig_orig, vec_orig=None):
        sig = np.asarray(sig_orig, dtype=np.float64).reshape(self.win_size, 10)
        if np.count_nonzero(sig[:, 9] != 0) < 2:      # bavoxel.hpp:47-52
            return
        self.plvec_voxels.append(sig)

    def pack(self):
        """CSR arrays of lvba_balm_create: the non-empty slots of every admitted voxel, ascending pose."""
        offs, idx, clu = [0], [], []
        for sig in self.plvec_voxels:
            nz = np.nonzero(sig[:, 9] != 0)[0]
            idx.append(nz.astype(np.int32))
            clu.append(sig[nz])
            offs.append(offs[-1] + len(nz))
        return (np.asarray(offs, np.int64), np.concatenate(idx) if idx else np.zeros(0, np.int32),
                np.concatenate(clu) if clu else np.zeros((0, 10)))


class BALM2:
    """bavoxel.hpp:587-767 on the GPU.  The packed device problem is cached per VOX_HESS instance."""

    def __init__(self, win_size=10, device=0):
        self.win_size = int(win_size)
        self.jac_leng = 6 * self.win_size
        self.device = device
        self._cache = (None, None)
        self.last_trace = []

    def _problem(self, voxhess):
        if self._cache[0] is not voxhess or self._cache[1] is None:
            off, idx, clu = voxhess.pack()
            if self._cache[1] is not None:
                self._cache[1].close()
            self._cache = (voxhess, BalmProblem(self.win_size, off, idx, clu, device=self.device))
        return self._cache[1]

    def divide_thread(self, x_stats, voxhess, x_ab=None):
        """-> (residual averaged over voxels, Hess [6N,6N], JacT [6N])   bavoxel.hpp:597-639"""
        H, g, c = self._problem(voxhess).eval(pack_x_stats(x_stats))
        return c, H, g

    def only_residual(self, x_stats, voxhess, x_ab=None, is_avg=False):
        """bavoxel.hpp:641-648"""
        return self._problem(voxhess).cost(pack_x_stats(x_stats), is_avg)

    def damping_iter(self, x_stats, voxhess):
        """Refines x_stats in place (bavoxel.hpp:662-767); the LM trace is kept in self.last_trace."""
        x, trace, rc = self._problem(voxhess).refine(pack_x_stats(x_stats))
        self.last_trace = trace
        unpack_x_stats(x, x_stats)
        return rc
