"""Host-side mirror of the reference's voxel front-end over the C-ABI.

    surf_map = VoxelMap(clouds, x_buf, voxel_size, eigen_ratio_array)   # cut_voxel per frame + recut per root
    voxhess  = surf_map.tras_opt()                                      # -> the VOX_HESS damping_iter takes
    plane, ok = surf_map.find_planes(X)                                 # recompute_local_planes

mirrors include/BALM/bavoxel.hpp:799-836 (cut_voxel), :391-464 (recut), :466-474 (tras_opt) and
src/lvba_system.cpp:1531-1565.  `Scans` keeps the clouds on the device between maps (window BA, stage 1, stage 2 and the
visual stage all re-cut the same clouds).  Everything runs in liblvba_hip.so on the GPU; this file packs arrays.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .balm import BalmProblem


def default_opts():
    o = L.VoxelOpts()
    L.load().lvba_voxel_default_opts(C.byref(o))
    return o


def _opts(voxel_size, eigen_ratio_array, min_points):
    o = default_opts()
    o.voxel_size = float(voxel_size)
    if eigen_ratio_array is not None:
        er = np.asarray(eigen_ratio_array, np.float32)
        for i in range(min(4, len(er))):
            o.eigen_ratio[i] = float(er[i])
    if min_points is not None:
        o.min_points = int(min_points)
    return o


class Scans:
    """A set of LiDAR clouds resident on a GPU (fp32 x, y, z per point)."""

    def __init__(self, clouds, device=0):
        """clouds: sequence of [n_i, >=3] float32 arrays (x, y, z first; row stride = the array's, so PCL-style padded
        points can be passed as they are)."""
        self.lib = L.load()
        n = len(clouds)
        keep = []
        ptrs = (C.c_void_p * max(n, 1))()
        counts = np.zeros(max(n, 1), np.int64)
        stride = None
        for f, c in enumerate(clouds):
            c = np.asarray(c)
            if c.dtype != np.float32:
                c = c.astype(np.float32)
            if c.ndim != 2 or c.shape[1] < 3:
                raise ValueError("each cloud must be [n, >=3] float32")
            if not c.flags["C_CONTIGUOUS"]:
                c = np.ascontiguousarray(c)
            if stride is None:
                stride = 4 * c.shape[1]
            elif stride != 4 * c.shape[1]:
                raise ValueError("all clouds must share one point stride")
            keep.append(c)
            ptrs[f] = c.ctypes.data if c.shape[0] else None
            counts[f] = c.shape[0]
        self.n_frames = n
        self.counts = counts[:n].copy()
        self._h = C.c_void_p()
        L.check(self.lib.lvba_scans_create(int(device), n, ptrs, counts, int(stride or 12), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.lvba_scans_destroy(self._h)
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

    @classmethod
    def _from_handle(cls, handle):
        self = cls.__new__(cls)
        self.lib = L.load()
        self._h = handle
        n = C.c_int32()
        L.check(self.lib.lvba_scans_info(self._h, C.byref(n), None))
        self.n_frames = n.value
        self.counts = np.zeros(max(self.n_frames, 1), np.int64)
        L.check(self.lib.lvba_scans_info(self._h, None, self.counts.ctypes.data))
        self.counts = self.counts[:self.n_frames]
        return self

    def download(self, frame):
        """Host copy [count, 3] fp32 of one frame."""
        out = np.zeros((int(self.counts[frame]), 3), np.float32)
        L.check(self.lib.lvba_scans_download(self._h, int(frame), out.reshape(-1) if out.size else np.zeros(1, np.float32)))
        return out

    def window_ba(self, poses, window_size=10, voxel_size=0.5, eigen_ratio_array=None, anchor_leaf=0.1, use_rel=True,
                  min_points=None, merge_only=False, lm_mode=0, **lm):
        """LvbaSystem::runWindowBA (src/lvba_system.cpp:204-310) on the resident scans.  Returns dict(anchor_poses,
        anchor_scans (a Scans), anchor_index, rel_poses, window_poses, windows)."""
        n = self.n_frames
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1)
        if poses.size != 12 * n:
            raise ValueError(f"poses must hold {n} x 12 doubles")
        o = L.WindowOpts()
        self.lib.lvba_window_default_opts(C.byref(o))
        o.window_size, o.use_rel, o.anchor_leaf = int(window_size), 1 if use_rel else 0, float(anchor_leaf)
        o.voxel = _opts(voxel_size, eigen_ratio_array, min_points)
        o.merge_only = 1 if merge_only else 0
        o.lm_mode = int(lm_mode)   # 0: all windows' LM in lock-step as one grouped problem, 1: one window at a time
        for k, v in lm.items():
            setattr(o.lm, k, v)
        nw = (n + o.window_size - 1) // o.window_size
        window_poses = np.zeros((n, 12))
        rel = np.zeros((n, 12))
        aidx = np.zeros(n, np.int32)
        aposes = np.zeros((max(nw, 1), 12))
        na = C.c_int32()
        h = C.c_void_p()
        info = (L.WindowInfo * max(nw, 1))()
        L.check(self.lib.lvba_window_ba(self._h, poses, C.byref(o), window_poses.ctypes.data, rel.reshape(-1), aidx,
                                        aposes.reshape(-1), C.byref(na), C.byref(h), info))
        return dict(anchor_poses=aposes[:na.value].copy(), anchor_scans=Scans._from_handle(h), anchor_index=aidx,
                    rel_poses=rel, window_poses=window_poses, windows=[info[i].as_dict() for i in range(nw)])

    @staticmethod
    def window_ba_multi(clouds, poses, devices, window_size=10, voxel_size=0.5, eigen_ratio_array=None, anchor_leaf=0.1, use_rel=True,
                        min_points=None, lm_mode=0, **lm):
        """The window stage over several GPUs (lvba_window_ba_multi): `clouds` (host arrays, one per frame) are dealt out to
        `devices` in contiguous runs of whole windows (lvba_window_split) -- a device id may repeat: several shares on one GPU --,
        every share runs on its own host thread, results come back in window order as from `window_ba`; the anchor scans live on
        devices[0]."""
        lib = L.load()
        n, D = len(clouds), len(devices)
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1)
        if poses.size != 12 * n:
            raise ValueError(f"poses must hold {n} x 12 doubles")
        fb = np.zeros(D + 1, np.int32)
        L.check(lib.lvba_window_split(n, int(window_size), D, fb))
        shares = [Scans(clouds[fb[k]:fb[k + 1]], device=int(devices[k])) for k in range(D) if fb[k + 1] > fb[k]]
        o = L.WindowOpts()
        lib.lvba_window_default_opts(C.byref(o))
        o.window_size, o.use_rel, o.anchor_leaf = int(window_size), 1 if use_rel else 0, float(anchor_leaf)
        o.voxel = _opts(voxel_size, eigen_ratio_array, min_points)
        o.lm_mode = int(lm_mode)
        for k, v in lm.items():
            setattr(o.lm, k, v)
        nw = (n + o.window_size - 1) // o.window_size
        window_poses, rel = np.zeros((n, 12)), np.zeros((n, 12))
        aidx = np.zeros(n, np.int32)
        aposes = np.zeros((max(nw, 1), 12))
        na, h = C.c_int32(), C.c_void_p()
        info = (L.WindowInfo * max(nw, 1))()
        hs = (C.c_void_p * len(shares))(*[sc._h.value for sc in shares])
        try:
            L.check(lib.lvba_window_ba_multi(len(shares), hs, poses, C.byref(o), window_poses.ctypes.data, rel.reshape(-1), aidx,
                                             aposes.reshape(-1), C.byref(na), C.byref(h), info))
        finally:
            for sc in shares:
                sc.close()
        return dict(anchor_poses=aposes[:na.value].copy(), anchor_scans=Scans._from_handle(h), anchor_index=aidx,
                    rel_poses=rel, window_poses=window_poses, windows=[info[i].as_dict() for i in range(nw)],
                    frame_begin=fb.copy())

    def lidar_ba(self, poses, window_enable=True, window_size=10, anchor_leaf=0.1, use_rel=True, stage1_enable=True,
                 stage_voxel_size=(0.5, 0.5), stage_eigen_ratio=((0.3, 0.1, 0.06, 0.03), (0.08, 0.08, 0.08, 0.08)),
                 window_eigen_ratio=None):
        """LvbaSystem::runLidarBA (src/lvba_system.cpp:312-410): window BA, global stage 1 / stage 2, pose composition.
        Returns (poses [n,12], report dict)."""
        n = self.n_frames
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1)
        if poses.size != 12 * n:
            raise ValueError(f"poses must hold {n} x 12 doubles")
        o = L.LidarBaOpts()
        self.lib.lvba_lidar_ba_default_opts(C.byref(o))
        o.window_enable, o.stage1_enable = int(bool(window_enable)), int(bool(stage1_enable))
        o.window.window_size, o.window.use_rel, o.window.anchor_leaf = int(window_size), int(bool(use_rel)), float(anchor_leaf)
        o.window.voxel = _opts(stage_voxel_size[0], window_eigen_ratio, None)   # stage1_root_voxel_size_, ratios in effect
        for i in range(2):
            o.stage_voxel_size[i] = float(stage_voxel_size[i])
            for k in range(4):
                o.stage_eigen_ratio[i][k] = float(stage_eigen_ratio[i][k])
        out = np.zeros(12 * n)
        rep = L.LidarBaReport()
        L.check(self.lib.lvba_lidar_ba(self._h, poses, C.byref(o), out, C.byref(rep)))
        return out.reshape(n, 12), rep.as_dict()

    @staticmethod
    def lidar_ba_multi(clouds, poses, devices, window_size=10, anchor_leaf=0.1, use_rel=True, stage1_enable=True,
                       stage_voxel_size=(0.5, 0.5), stage_eigen_ratio=((0.3, 0.1, 0.06, 0.03), (0.08, 0.08, 0.08, 0.08)),
                       window_eigen_ratio=None):
        """runLidarBA with the window stage over several GPUs (lvba_lidar_ba_multi): returns (poses [n,12], report dict)."""
        lib = L.load()
        n, D = len(clouds), len(devices)
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1)
        if poses.size != 12 * n:
            raise ValueError(f"poses must hold {n} x 12 doubles")
        fb = np.zeros(D + 1, np.int32)
        L.check(lib.lvba_window_split(n, int(window_size), D, fb))
        shares = [Scans(clouds[fb[k]:fb[k + 1]], device=int(devices[k])) for k in range(D) if fb[k + 1] > fb[k]]
        o = L.LidarBaOpts()
        lib.lvba_lidar_ba_default_opts(C.byref(o))
        o.window_enable, o.stage1_enable = 1, int(bool(stage1_enable))
        o.window.window_size, o.window.use_rel, o.window.anchor_leaf = int(window_size), int(bool(use_rel)), float(anchor_leaf)
        o.window.voxel = _opts(stage_voxel_size[0], window_eigen_ratio, None)
        for i in range(2):
            o.stage_voxel_size[i] = float(stage_voxel_size[i])
            for k in range(4):
                o.stage_eigen_ratio[i][k] = float(stage_eigen_ratio[i][k])
        out = np.zeros(12 * n)
        rep = L.LidarBaReport()
        hs = (C.c_void_p * len(shares))(*[sc._h.value for sc in shares])
        try:
            L.check(lib.lvba_lidar_ba_multi(len(shares), hs, poses, C.byref(o), out, C.byref(rep)))
        finally:
            for sc in shares:
                sc.close()
        return out.reshape(n, 12), rep.as_dict()

    def voxel_map(self, poses, voxel_size=1.0, eigen_ratio_array=None, min_points=None, frame_begin=0, n_frames=None):
        """Map of frames [frame_begin, frame_begin + n_frames) at `poses` [n_frames, 12]."""
        n = self.n_frames - frame_begin if n_frames is None else int(n_frames)
        return VoxelMap(None, poses, voxel_size, eigen_ratio_array, min_points, _scans=(self, int(frame_begin), n))


class VoxelMap:
    """Adaptive-voxel plane map of a window of scans, resident on a GPU."""

    def __init__(self, clouds, poses, voxel_size=1.0, eigen_ratio_array=None, min_points=None, device=0, _scans=None):
        self.lib = L.load()
        o = _opts(voxel_size, eigen_ratio_array, min_points)
        self.voxel_size = float(voxel_size)
        self._h = C.c_void_p()
        if _scans is not None:
            sc, begin, n = _scans
            poses = np.ascontiguousarray(poses, np.float64).reshape(-1)
            if poses.size != 12 * n:
                raise ValueError(f"poses must hold {n} x 12 doubles")
            L.check(self.lib.lvba_voxmap_build_scans(sc._h, begin, n, poses, C.byref(o), C.byref(self._h)))
        else:
            n = len(clouds)
            poses = np.ascontiguousarray(poses, np.float64).reshape(-1)
            if poses.size != 12 * n:
                raise ValueError(f"poses must hold {n} x 12 doubles")
            keep, ptrs, counts, stride = [], (C.c_void_p * max(n, 1))(), np.zeros(max(n, 1), np.int64), None
            for f, c in enumerate(clouds):
                c = np.asarray(c)
                if c.dtype != np.float32:
                    c = c.astype(np.float32)
                if c.ndim != 2 or c.shape[1] < 3:
                    raise ValueError("each cloud must be [n, >=3] float32")
                if not c.flags["C_CONTIGUOUS"]:
                    c = np.ascontiguousarray(c)
                if stride is None:
                    stride = 4 * c.shape[1]
                elif stride != 4 * c.shape[1]:
                    raise ValueError("all clouds must share one point stride")
                keep.append(c)
                ptrs[f] = c.ctypes.data if c.shape[0] else None
                counts[f] = c.shape[0]
            L.check(self.lib.lvba_voxmap_build(int(device), n, ptrs, counts, int(stride or 12), poses, C.byref(o),
                                               C.byref(self._h)))
        self.n_frames = n
        info = L.VoxmapInfo()
        L.check(self.lib.lvba_voxmap_info(self._h, C.byref(info)))
        self.info = {f: getattr(info, f) for f, _ in info._fields_}

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.lvba_voxmap_destroy(self._h)
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

    def export(self):
        """(voxel_off [V+1], pose_idx [F], clusters [F,10], voxel_key [V,4]) of the admitted voxels."""
        V, F = self.info["n_voxels"], self.info["n_factors"]
        off = np.zeros(V + 1, np.int64)
        idx = np.zeros(F, np.int32)
        cl = np.zeros((F, 10))
        key = np.zeros((V, 4), np.int64)
        L.check(self.lib.lvba_voxmap_export(self._h, off.ctypes.data, idx.ctypes.data, cl.ctypes.data, key.ctypes.data))
        return off, idx, cl, key

    def tras_opt(self, ordering=None, band_frac=None):
        """The packed problem of all admitted plane voxels (every root's tras_opt, bavoxel.hpp:466-474)."""
        h = C.c_void_p()
        L.check(self.lib.lvba_voxmap_to_balm(self._h, C.byref(h)))
        return BalmProblem._from_handle(h, self.n_frames, self.info["n_voxels"], ordering, band_frac)

    def find_planes(self, X):
        """(plane [n,4] = unit normal and d, valid [n]) for world points X [n,3] (src/lvba_system.cpp:1531-1565)."""
        X = np.ascontiguousarray(X, np.float64).reshape(-1, 3)
        plane = np.zeros((len(X), 4))
        valid = np.zeros(len(X), np.uint8)
        L.check(self.lib.lvba_voxmap_find_planes(self._h, len(X), X.reshape(-1), plane.reshape(-1), valid))
        return plane, valid
