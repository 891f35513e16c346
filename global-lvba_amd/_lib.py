"""ctypes binding of liblvba_hip.so (include/lvba_hip.h).  No torch types cross this boundary."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LVBA_HIP_LIB") or os.path.join(HERE, "liblvba_hip.so")  # override: A/B runs of two builds

# every extern "C" symbol include/lvba_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "lvba_version", "lvba_last_error", "lvba_device_count", "lvba_balm_default_opts", "lvba_shard_range",
    "lvba_balm_create", "lvba_balm_create_dev", "lvba_balm_destroy", "lvba_balm_configure", "lvba_balm_info", "lvba_balm_cost",
    "lvba_balm_eval", "lvba_balm_eval_blocks", "lvba_balm_solve", "lvba_balm_refine", "lvba_balm_lm_begin", "lvba_balm_lm_step",
    "lvba_balm_lm_end", "lvba_balm_set_groups", "lvba_balm_refine_groups", "lvba_balm_set_profiling", "lvba_balm_get_profile", "lvba_balm_get_ordering", "lvba_balm_nd_model",
    "lvba_dist_unique_id", "lvba_balm_dist_init", "lvba_balm_dist_init_external", "lvba_visual_dist_init_external",
    "lvba_visual_default_opts", "lvba_visual_create", "lvba_visual_destroy", "lvba_visual_cost", "lvba_visual_linearize", "lvba_visual_info", "lvba_visual_dist_init",
    "lvba_visual_refine",
    "lvba_voxel_default_opts", "lvba_voxmap_build", "lvba_voxmap_destroy", "lvba_voxmap_info", "lvba_voxmap_export",
    "lvba_voxmap_to_balm", "lvba_voxmap_find_planes", "lvba_scans_create", "lvba_scans_destroy", "lvba_voxmap_build_scans",
    "lvba_release_cached_memory", "lvba_window_default_opts", "lvba_window_ba", "lvba_window_split", "lvba_window_ba_multi", "lvba_scans_info", "lvba_scans_download",
    "lvba_lidar_ba_default_opts", "lvba_lidar_ba", "lvba_lidar_ba_multi", "lvba_triangulate_tracks",
    "lvba_depth_render", "lvba_depth_upload", "lvba_depth_info", "lvba_depth_download", "lvba_depth_destroy",
    "lvba_fuse_default_opts", "lvba_fuse_tracks",
]

OK, ERR_ARG, ERR_DEVICE, ERR_NOMEM, ERR_UNSUPPORTED, ERR_DIST, ERR_STATE = 0, -1, -2, -3, -4, -5, -6
NUM_FACTORIZATION, NUM_NONFINITE = 1, 2


class BalmOpts(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("reserved", C.c_int32), ("u0", C.c_double), ("v0", C.c_double),
                ("rel_tol", C.c_double)]


class LmTrace(C.Structure):
    _fields_ = [("iter", C.c_int32), ("accepted", C.c_int32), ("evaluated", C.c_int32), ("status", C.c_int32),
                ("residual1", C.c_double), ("residual2", C.c_double), ("u", C.c_double), ("v", C.c_double),
                ("q", C.c_double), ("q1", C.c_double)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


class BalmInfo(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("n_ranks", C.c_int32), ("n_voxels", C.c_int64),
                ("n_voxels_global", C.c_int64), ("n_factors", C.c_int64), ("n_pairs", C.c_int64),
                ("n_chunks", C.c_int64), ("n_blocks", C.c_int64), ("band_blocks", C.c_int32), ("use_band", C.c_int32),
                ("hess_bytes", C.c_int64), ("device_bytes", C.c_int64), ("allreduce_bytes", C.c_int64),
                ("twist_panels", C.c_int32), ("solve_ranks", C.c_int32), ("trial_linearised", C.c_int32), ("y_fp32", C.c_int32),
                ("nd_kind", C.c_int32), ("nd_arcs", C.c_int32), ("nd_sep_poses", C.c_int32), ("nd_sep_band_blocks", C.c_int32),
                ("nd_model_band_ms", C.c_double), ("nd_model_nd_ms", C.c_double)]


class NdModel(C.Structure):
    _fields_ = [("n_ranks", C.c_int32), ("arcs", C.c_int32), ("sep_poses", C.c_int32), ("sep_band_blocks", C.c_int32),
                ("max_arc_poses", C.c_int32), ("max_arc_band_blocks", C.c_int32), ("band_ms", C.c_double), ("nd_ms", C.c_double)]


class Prof(C.Structure):
    _fields_ = [("cost_ms", C.c_double), ("cost_calls", C.c_int64), ("eval_ms", C.c_double),
                ("eval_calls", C.c_int64), ("solve_ms", C.c_double), ("solve_calls", C.c_int64),
                ("reduce_ms", C.c_double), ("reduce_calls", C.c_int64), ("cost_kernel_ms", C.c_double),
                ("eval_kernel_ms", C.c_double)]


class VisualOpts(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("reserved", C.c_int32), ("initial_radius", C.c_double), ("max_radius", C.c_double),
                ("min_radius", C.c_double), ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double),
                ("max_lm_diagonal", C.c_double), ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
                ("parameter_tolerance", C.c_double)]


class VisualTrace(C.Structure):
    _fields_ = [("iter", C.c_int32), ("accepted", C.c_int32), ("valid", C.c_int32), ("reserved", C.c_int32),
                ("cost", C.c_double), ("cost_change", C.c_double), ("step_norm", C.c_double), ("radius", C.c_double),
                ("rho", C.c_double), ("gradient_max_norm", C.c_double)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_ if f != "reserved"}


class FuseOpts(C.Structure):
    _fields_ = [("obser_thr", C.c_int32), ("reserved", C.c_int32), ("min_view_angle_deg", C.c_double),
                ("reproj_mean_thr_px", C.c_double)]


class VoxelOpts(C.Structure):
    _fields_ = [("voxel_size", C.c_double), ("eigen_ratio", C.c_float * 4), ("min_points", C.c_int32),
                ("layer_limit", C.c_int32)]


class VoxmapInfo(C.Structure):
    _fields_ = [("n_points", C.c_int64), ("n_roots", C.c_int64), ("n_planes", C.c_int64), ("n_voxels", C.c_int64),
                ("n_factors", C.c_int64), ("upload_ms", C.c_double), ("key_ms", C.c_double), ("sort_ms", C.c_double),
                ("count_ms", C.c_double), ("write_ms", C.c_double)]


class WindowOpts(C.Structure):
    _fields_ = [("window_size", C.c_int32), ("use_rel", C.c_int32), ("anchor_leaf", C.c_double), ("voxel", VoxelOpts),
                ("lm", BalmOpts), ("merge_only", C.c_int32), ("lm_mode", C.c_int32)]


class WindowInfo(C.Structure):
    _fields_ = [("start", C.c_int32), ("n_frames", C.c_int32), ("skipped", C.c_int32), ("anchor", C.c_int32),
                ("n_iter", C.c_int32), ("lm_status", C.c_int32), ("n_voxels", C.c_int64), ("n_factors", C.c_int64),
                ("n_anchor_points", C.c_int64), ("cost_first", C.c_double), ("cost_last", C.c_double), ("map_ms", C.c_double),
                ("solve_ms", C.c_double), ("merge_ms", C.c_double), ("setup_ms", C.c_double)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


class LidarBaOpts(C.Structure):
    _fields_ = [("window", WindowOpts), ("window_enable", C.c_int32), ("stage1_enable", C.c_int32),
                ("stage_voxel_size", C.c_double * 2), ("stage_eigen_ratio", (C.c_float * 4) * 2), ("lm", BalmOpts)]


class LidarBaReport(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("n_windows", C.c_int32), ("n_windows_skipped", C.c_int32), ("n_anchors", C.c_int32),
                ("stage_ran", C.c_int32 * 2), ("stage_iters", C.c_int32 * 2), ("stage_status", C.c_int32 * 2),
                ("reserved", C.c_int32), ("stage_voxels", C.c_int64 * 2), ("stage_factors", C.c_int64 * 2),
                ("stage_cost_first", C.c_double * 2), ("stage_cost_last", C.c_double * 2), ("window_ms", C.c_double),
                ("stage_ms", C.c_double * 2)]

    def as_dict(self):
        out = {}
        for f, t in self._fields_:
            v = getattr(self, f)
            out[f] = list(v) if hasattr(v, "__len__") else v
        return out


TERMINATION = {0: "NO_CONVERGENCE", 1: "CONVERGENCE(function)", 2: "CONVERGENCE(parameter)", 3: "CONVERGENCE(gradient)",
               4: "CONVERGENCE(radius)", 5: "FAILURE"}


class LvbaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"lvba status {code}: {msg}")
        self.code = code


_lib = None


def load():
    """Load the HIP library.  Fails loudly if it has not been built (python __graft_entry__.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the hot path)")
    lib = C.CDLL(LIB_PATH)
    f64p = np.ctypeslib.ndpointer(np.float64, flags="C")
    i64p = np.ctypeslib.ndpointer(np.int64, flags="C")
    i32p = np.ctypeslib.ndpointer(np.int32, flags="C")
    H = C.c_void_p
    lib.lvba_version.restype = C.c_int32
    lib.lvba_last_error.restype = C.c_char_p
    lib.lvba_device_count.restype = C.c_int32
    lib.lvba_balm_default_opts.argtypes = [C.POINTER(BalmOpts)]
    lib.lvba_balm_default_opts.restype = None
    lib.lvba_shard_range.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.lvba_shard_range.restype = None
    lib.lvba_balm_create.argtypes = [C.c_int32, C.c_int64, i64p, i32p, f64p, C.c_int32, C.POINTER(H)]
    lib.lvba_balm_create_dev.argtypes = [C.c_int32, C.c_int64, i64p, i32p, C.c_void_p, C.c_int32, C.POINTER(H)]
    lib.lvba_balm_destroy.argtypes = [H]
    lib.lvba_balm_configure.argtypes = [H, C.c_int32, C.c_double]
    lib.lvba_balm_info.argtypes = [H, C.POINTER(BalmInfo)]
    lib.lvba_balm_cost.argtypes = [H, f64p, C.c_int32, C.POINTER(C.c_double)]
    lib.lvba_balm_eval.argtypes = [H, f64p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    lib.lvba_balm_eval_blocks.argtypes = [H, f64p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p,
                                          C.POINTER(C.c_double)]
    lib.lvba_balm_solve.argtypes = [H, C.c_double, f64p]
    lib.lvba_balm_refine.argtypes = [H, f64p, C.POINTER(BalmOpts), C.POINTER(LmTrace), C.POINTER(C.c_int32)]
    lib.lvba_balm_lm_begin.argtypes = [H, f64p, C.POINTER(BalmOpts)]
    lib.lvba_balm_lm_step.argtypes = [H, C.POINTER(LmTrace), C.POINTER(C.c_int32)]
    lib.lvba_balm_lm_end.argtypes = [H, C.c_void_p]
    lib.lvba_balm_set_groups.argtypes = [H, C.c_int32, i32p, i64p]
    lib.lvba_balm_refine_groups.argtypes = [H, f64p, C.POINTER(BalmOpts), i32p, i32p, f64p, f64p]
    lib.lvba_balm_set_profiling.argtypes = [H, C.c_int32]
    lib.lvba_balm_get_profile.argtypes = [H, C.POINTER(Prof), C.c_int32]
    lib.lvba_balm_get_ordering.argtypes = [H, i32p]
    lib.lvba_balm_nd_model.argtypes = [H, C.c_int32, C.POINTER(NdModel)]
    lib.lvba_dist_unique_id.argtypes = [C.c_char_p]
    lib.lvba_balm_dist_init_external.argtypes = [H, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.lvba_visual_dist_init_external.argtypes = [H, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.lvba_balm_dist_init.argtypes = [H, C.c_int32, C.c_int32, C.c_char_p]
    u8p = np.ctypeslib.ndpointer(np.uint8, flags="C")
    lib.lvba_visual_default_opts.argtypes = [C.POINTER(VisualOpts)]
    lib.lvba_visual_default_opts.restype = None
    lib.lvba_visual_create.argtypes = [C.c_int32, C.c_int64, i64p, C.c_void_p, C.c_void_p, f64p, u8p, f64p, C.c_double,
                                       C.c_double, C.c_int32, C.POINTER(H)]
    lib.lvba_visual_destroy.argtypes = [H]
    lib.lvba_visual_cost.argtypes = [H, f64p, f64p, f64p, C.POINTER(C.c_double)]
    lib.lvba_visual_info.argtypes = [H, C.POINTER(BalmInfo)]
    lib.lvba_visual_dist_init.argtypes = [H, C.c_int32, C.c_int32, C.c_char_p]
    lib.lvba_visual_linearize.argtypes = [H, f64p, f64p, f64p, C.c_double, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    lib.lvba_visual_refine.argtypes = [H, f64p, f64p, f64p, C.POINTER(VisualOpts), C.POINTER(VisualTrace), C.c_int32,
                                       C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.lvba_voxel_default_opts.argtypes = [C.POINTER(VoxelOpts)]
    lib.lvba_voxel_default_opts.restype = None
    lib.lvba_voxmap_build.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_void_p), i64p, C.c_int32, f64p,
                                      C.POINTER(VoxelOpts), C.POINTER(H)]
    lib.lvba_voxmap_destroy.argtypes = [H]
    lib.lvba_scans_create.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_void_p), i64p, C.c_int32, C.POINTER(H)]
    lib.lvba_scans_destroy.argtypes = [H]
    lib.lvba_voxmap_build_scans.argtypes = [H, C.c_int32, C.c_int32, f64p, C.POINTER(VoxelOpts), C.POINTER(H)]
    lib.lvba_voxmap_info.argtypes = [H, C.POINTER(VoxmapInfo)]
    lib.lvba_voxmap_export.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.lvba_voxmap_to_balm.argtypes = [H, C.POINTER(H)]
    lib.lvba_voxmap_find_planes.argtypes = [H, C.c_int64, f64p, f64p, u8p]
    lib.lvba_release_cached_memory.restype = C.c_int64
    lib.lvba_window_default_opts.argtypes = [C.POINTER(WindowOpts)]
    lib.lvba_window_default_opts.restype = None
    lib.lvba_window_ba.argtypes = [H, f64p, C.POINTER(WindowOpts), C.c_void_p, f64p, i32p, f64p, C.POINTER(C.c_int32),
                                   C.POINTER(H), C.POINTER(WindowInfo)]
    lib.lvba_window_split.argtypes = [C.c_int32, C.c_int32, C.c_int32, i32p]
    lib.lvba_window_ba_multi.argtypes = [C.c_int32, C.POINTER(H), f64p, C.POINTER(WindowOpts), C.c_void_p, f64p, i32p, f64p,
                                         C.POINTER(C.c_int32), C.POINTER(H), C.POINTER(WindowInfo)]
    lib.lvba_lidar_ba_default_opts.argtypes = [C.POINTER(LidarBaOpts)]
    lib.lvba_lidar_ba_default_opts.restype = None
    lib.lvba_lidar_ba.argtypes = [H, f64p, C.POINTER(LidarBaOpts), f64p, C.POINTER(LidarBaReport)]
    lib.lvba_lidar_ba_multi.argtypes = [C.c_int32, C.POINTER(H), f64p, C.POINTER(LidarBaOpts), f64p, C.POINTER(LidarBaReport)]
    lib.lvba_triangulate_tracks.argtypes = [C.c_int32, C.c_int32, C.c_int64, i64p, C.c_void_p, C.c_void_p, f64p, f64p, f64p, f64p,
                                            f64p, i32p, u8p]
    lib.lvba_depth_render.argtypes = [C.c_void_p, f64p, f64p, C.c_int32, f64p, f64p, f64p, f64p, C.c_int32, C.c_int32, C.c_double,
                                      C.c_double, C.POINTER(C.c_void_p)]
    lib.lvba_depth_upload.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, np.ctypeslib.ndpointer(np.float32, flags="C"),
                                      C.POINTER(C.c_void_p)]
    lib.lvba_depth_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.lvba_depth_download.argtypes = [C.c_void_p, C.c_int32, np.ctypeslib.ndpointer(np.float32, flags="C")]
    lib.lvba_depth_destroy.argtypes = [C.c_void_p]
    lib.lvba_depth_destroy.restype = None
    lib.lvba_fuse_default_opts.argtypes = [C.POINTER(FuseOpts)]
    lib.lvba_fuse_default_opts.restype = None
    lib.lvba_fuse_tracks.argtypes = [C.c_int32, C.c_void_p, C.c_int32, f64p, f64p, f64p, C.c_int64, i64p, C.c_void_p, C.c_void_p,
                                     C.POINTER(FuseOpts), C.c_void_p, f64p, f64p, C.c_void_p]
    lib.lvba_scans_info.argtypes = [H, C.POINTER(C.c_int32), C.c_void_p]
    lib.lvba_scans_download.argtypes = [H, C.c_int32, np.ctypeslib.ndpointer(np.float32, flags="C")]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int:  # default
            fn.restype = C.c_int32
    _lib = lib
    return lib


def check(rc, allow_numeric=False):
    if rc == OK or (allow_numeric and rc > 0):
        return rc
    raise LvbaError(rc, load().lvba_last_error().decode(errors="replace"))
