"""ctypes wrapper of oracle/_ref/liblvba_system_ref.so -- the REFERENCE'S OWN src/lvba_system.cpp + src/dataset_io.cpp compiled
against the stand-ins of oracle/shim (see oracle/ref_glue_system.cpp).  TEST INFRASTRUCTURE ONLY: imported by tests/ and
tests/golden/make_golden.py, never by the product.

`ReferenceSystem(dataset_dir, params)` runs the reference's constructor (parameters + DatasetIO load); the methods are the
stages of LvbaSystem::runFullPipeline, callable one at a time so that each can be compared with its restatement."""
import ctypes

import numpy as np

from . import build_dropin, ref_system_path

_LIB = None
_LIBS = {}


def load(dropin=False):
    """dropin=True: oracle/_ref/liblvba_system_dropin.so -- the same reference sources with BALM2::damping_iter and ceres::Solve
    going to the product's liblvba_hip.so (GPU) through include/lvba_adapter.hpp."""
    global _LIB
    if dropin not in _LIBS:
        so = build_dropin() if dropin else ref_system_path()
        if so is None:
            _LIBS[dropin] = None
            return None
        lib = ctypes.CDLL(so)
        vp, c_int, c_dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
        f64p = np.ctypeslib.ndpointer(np.float64, flags="C")
        f32p = np.ctypeslib.ndpointer(np.float32, flags="C")
        i32p = np.ctypeslib.ndpointer(np.int32, flags="C")
        lib.ref_sys_set_param.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        lib.ref_sys_create.restype = vp
        lib.ref_sys_destroy.argtypes = [vp]
        for name in ("n_scans", "n_clouds", "n_images", "init", "run_lidar_ba", "build_grid_map", "update_camera_poses",
                     "generate_depth", "build_tracks", "n_tracks"):
            fn = getattr(lib, "ref_sys_" + name)
            fn.restype, fn.argtypes = c_int, [vp]
        lib.ref_sys_scan_poses.argtypes = [vp, c_int, f64p, f64p, f64p]
        lib.ref_sys_set_scan_poses.argtypes = [vp, f64p, f64p]
        lib.ref_sys_cloud_size.restype, lib.ref_sys_cloud_size.argtypes = c_int, [vp, c_int]
        lib.ref_sys_cloud.argtypes = [vp, c_int, f32p]
        lib.ref_sys_image_ids.argtypes = [vp, f64p]
        lib.ref_sys_image_poses.argtypes = [vp, c_int, f64p, f64p]
        lib.ref_sys_camera.argtypes = [vp, f64p]
        lib.ref_sys_extrinsics.argtypes = [vp, f64p, f64p]
        lib.ref_sys_anchor_index.argtypes = [vp, i32p]
        lib.ref_sys_rel_poses.argtypes = [vp, f64p, f64p]
        lib.ref_sys_grid_points.restype, lib.ref_sys_grid_points.argtypes = ctypes.c_int64, [vp]
        lib.ref_sys_n_voxel_ids.restype, lib.ref_sys_n_voxel_ids.argtypes = c_int, [vp, c_int]
        lib.ref_sys_depth.argtypes = [vp, c_int, f32p]
        lib.ref_sys_cam_poses.argtypes = [vp, c_int, f64p, f64p]
        lib.ref_sys_set_keypoints.argtypes = [vp, c_int, c_int, f32p]
        lib.ref_sys_set_matches.argtypes = [vp, c_int, c_int, c_int, i32p]
        lib.ref_sys_set_fusion_params.argtypes = [vp, c_int, c_dbl, c_dbl]
        lib.ref_sys_track_sizes.restype, lib.ref_sys_track_sizes.argtypes = c_int, [vp, c_int, ctypes.POINTER(c_int)]
        lib.ref_sys_track.argtypes = [vp, c_int, f64p, i32p, i32p]
        lib.ref_sys_load_colmap_db.restype, lib.ref_sys_load_colmap_db.argtypes = c_int, [vp]
        lib.ref_sys_n_keypoints.restype, lib.ref_sys_n_keypoints.argtypes = c_int, [vp, c_int]
        lib.ref_sys_keypoints.argtypes = [vp, c_int, f32p]
        lib.ref_sys_n_matches.restype, lib.ref_sys_n_matches.argtypes = c_int, [vp, c_int, c_int]
        lib.ref_sys_matches.argtypes = [vp, c_int, c_int, i32p]
        lib.ref_sys_export_colmap.restype, lib.ref_sys_export_colmap.argtypes = c_int, [vp, c_int, c_int]
        lib.ref_sys_optimize.restype = c_int
        lib.ref_sys_optimize.argtypes = [vp, vp, vp, vp, c_int, c_int]
        lib.ref_sys_problem_info.argtypes = [i32p]
        lib.ref_sys_problem_cost.restype = c_dbl
        lib.ref_sys_problem_blocks.argtypes = [f64p, f64p, f64p, f64p, i32p, i32p, i32p]
        lib.ref_sys_problem_residuals.argtypes = [i32p, i32p, i32p, f64p, f64p]
        lib.ref_sys_problem_uv.argtypes = [f64p]
        if dropin:
            lib.ref_sys_optimize_dropin.restype, lib.ref_sys_optimize_dropin.argtypes = c_int, [vp]
            lib.ref_sys_dropin_stats.argtypes = [i32p, f64p]
            lib.ref_sys_dropin_call_diffs.argtypes = [f64p]
        _LIBS[dropin] = lib
    return _LIBS[dropin]


def available(dropin=False):
    return load(dropin) is not None


def _fmt(v):
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, (list, tuple, np.ndarray)):
        return " ".join(repr(float(x)) for x in np.asarray(v).reshape(-1))
    return repr(v) if isinstance(v, float) else str(v)


class ReferenceSystem:
    """One lvba::LvbaSystem.  `params`: the ROS parameter names of the reference's launch/config files (e.g.
    "cam_model/cam_fx", "window_ba/size", "extrin_calib/Rcl") -> values; "data_config/data_path" is set from `dataset_dir`."""

    def __init__(self, dataset_dir, params=None, dropin=False):
        self.lib = load(dropin)
        self.dropin = dropin
        if self.lib is None:
            raise RuntimeError("oracle/_ref/liblvba_system_%s.so is not available (no reference sources, no prebuilt library)"
                               % ("dropin" if dropin else "ref"))
        self.lib.ref_sys_clear_params()
        p = dict(params or {})
        d = str(dataset_dir)
        p["data_config/data_path"] = d if d.endswith("/") else d + "/"
        for k, v in p.items():
            self.lib.ref_sys_set_param(k.encode(), _fmt(v).encode())
        self.h = self.lib.ref_sys_create()
        if not self.h:
            raise RuntimeError("the reference's LvbaSystem constructor threw")
        self.n_scans = self.lib.ref_sys_n_scans(self.h)
        self.n_clouds = self.lib.ref_sys_n_clouds(self.h)
        self.n_images = self.lib.ref_sys_n_images(self.h)

    def close(self):
        if self.h:
            self.lib.ref_sys_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    # ---- DatasetIO -----------------------------------------------------------------------------------------------------
    def scan_poses(self, before=False):
        n = self.n_scans
        R, p, t = np.zeros((n, 3, 3)), np.zeros((n, 3)), np.zeros(n)
        self.lib.ref_sys_scan_poses(self.h, int(before), R.reshape(-1), p.reshape(-1), t)
        return R, p, t

    def set_scan_poses(self, R, p):
        self.lib.ref_sys_set_scan_poses(self.h, np.ascontiguousarray(R, np.float64).reshape(-1), np.ascontiguousarray(p, np.float64).reshape(-1))

    def cloud(self, i):
        n = self.lib.ref_sys_cloud_size(self.h, i)
        out = np.zeros((n, 4), np.float32)
        self.lib.ref_sys_cloud(self.h, i, out.reshape(-1))
        return out

    def image_ids(self):
        out = np.zeros(self.n_images)
        self.lib.ref_sys_image_ids(self.h, out)
        return out

    def image_poses(self, updated=False):
        n = self.n_images
        R, t = np.zeros((n, 3, 3)), np.zeros((n, 3))
        self.lib.ref_sys_image_poses(self.h, int(updated), R.reshape(-1), t.reshape(-1))
        return R, t

    def camera(self):
        out = np.zeros(11)
        self.lib.ref_sys_camera(self.h, out)
        return dict(width=int(out[0]), height=int(out[1]), intr=out[2:10].copy(), scale=out[10])

    # ---- stages --------------------------------------------------------------------------------------------------------
    def _call(self, name):
        rc = getattr(self.lib, "ref_sys_" + name)(self.h)
        if rc != 0:
            raise RuntimeError(f"reference {name} failed ({rc})")

    def init(self):
        self._call("init")
        Rci, tci = np.zeros((3, 3)), np.zeros(3)
        self.lib.ref_sys_extrinsics(self.h, Rci.reshape(-1), tci)
        return Rci, tci

    def run_lidar_ba(self):
        """runLidarBA: returns (anchor index per scan, relative pose of every scan to its anchor (R, p))."""
        self._call("run_lidar_ba")
        n = self.n_scans
        idx, R, p = np.zeros(n, np.int32), np.zeros((n, 3, 3)), np.zeros((n, 3))
        self.lib.ref_sys_anchor_index(self.h, idx)
        self.lib.ref_sys_rel_poses(self.h, R.reshape(-1), p.reshape(-1))
        return idx, R, p

    def build_grid_map(self):
        self._call("build_grid_map")
        return int(self.lib.ref_sys_grid_points(self.h)), [self.lib.ref_sys_n_voxel_ids(self.h, i) for i in range(self.n_images)]

    def update_camera_poses(self):
        self._call("update_camera_poses")
        return self.image_poses(updated=True)

    def generate_depth(self, width, height):
        self._call("generate_depth")
        out = np.zeros((self.n_images, height, width), np.float32)
        for i in range(self.n_images):
            self.lib.ref_sys_depth(self.h, i, out[i].reshape(-1))
        return out

    def cam_poses(self, optimized=True):
        n = self.n_images
        R, t = np.zeros((n, 3, 3)), np.zeros((n, 3))
        self.lib.ref_sys_cam_poses(self.h, int(optimized), R.reshape(-1), t.reshape(-1))
        return R, t

    def set_features(self, keypoints, matches):
        """keypoints: list over images of [n, 2] pixel arrays; matches: {(i, j): [m, 2] key point index pairs}, i < j."""
        for i, kp in enumerate(keypoints):
            kp = np.ascontiguousarray(kp, np.float32)
            self.lib.ref_sys_set_keypoints(self.h, i, len(kp), kp.reshape(-1))
        for (i, j), m in matches.items():
            m = np.ascontiguousarray(m, np.int32)
            self.lib.ref_sys_set_matches(self.h, i, j, len(m), m.reshape(-1))

    def build_tracks(self, obser_thr=3, min_view_angle_deg=8.0, reproj_mean_thr_px=3.0):
        """BuildTracksAndFuse3D: list of dict(X, obs [n,2] (image, key point), inliers [k])."""
        self.lib.ref_sys_set_fusion_params(self.h, obser_thr, min_view_angle_deg, reproj_mean_thr_px)
        self._call("build_tracks")
        out = []
        for t in range(self.lib.ref_sys_n_tracks(self.h)):
            n_inl = ctypes.c_int()
            n = self.lib.ref_sys_track_sizes(self.h, t, ctypes.byref(n_inl))
            X, obs, inl = np.zeros(3), np.zeros((n, 2), np.int32), np.zeros(n_inl.value, np.int32)
            self.lib.ref_sys_track(self.h, t, X, obs.reshape(-1), inl)
            out.append(dict(X=X, obs=obs, inliers=inl))
        return out

    def load_colmap_db(self):
        """loadFromColmapDB: (accepted, key points per image [n,4] = x y sigma extremum, {(i, j): matches [m,2]} for all i < j)."""
        rc = self.lib.ref_sys_load_colmap_db(self.h)
        if rc < 0:
            raise RuntimeError("reference loadFromColmapDB threw")
        if rc == 0:
            return False, [], {}
        kps = []
        for i in range(self.n_images):
            k = np.zeros((self.lib.ref_sys_n_keypoints(self.h, i), 4), np.float32)
            self.lib.ref_sys_keypoints(self.h, i, k.reshape(-1))
            kps.append(k)
        matches = {}
        for i in range(self.n_images):
            for j in range(i + 1, self.n_images):
                m = np.zeros((self.lib.ref_sys_n_matches(self.h, i, j), 2), np.int32)
                self.lib.ref_sys_matches(self.h, i, j, m.reshape(-1))
                matches[(i, j)] = m
        return True, kps, matches

    def export_colmap(self, width, height):
        """VisualizeOptComparison: writes <dataset>/Colmap/sparse/{images,points3D}.txt (images are a synthetic pattern: b = x,
        g = y, r = x + y mod 256).  Empties the reference's clouds: call it last."""
        if self.lib.ref_sys_export_colmap(self.h, int(width), int(height)) != 0:
            raise RuntimeError("reference VisualizeOptComparison threw")

    def optimize_dropin(self):
        """optimizeCameraPoses of the drop-in build: the reference assembles its ceres::Problem, ceres::Solve runs
        lvba_visual_refine on the GPU, the reference writes the result back."""
        if self.lib.ref_sys_optimize_dropin(self.h) != 0:
            raise RuntimeError("reference optimizeCameraPoses (drop-in) threw")
        return self.dropin_stats()

    def dropin_stats(self):
        out, cost = np.zeros(5, np.int32), np.zeros(2)
        self.lib.ref_sys_dropin_stats(out, cost)
        diffs = np.zeros(16)
        self.lib.ref_sys_dropin_call_diffs(diffs)
        return dict(lidar_calls=int(out[0]), lidar_iterations=int(out[1]), visual_calls=int(out[2]), visual_termination=int(out[3]),
                    visual_iterations=int(out[4]), visual_cost0=float(cost[0]), visual_cost1=float(cost[1]),
                    lidar_call_diffs=diffs[:max(0, min(16, int(out[0])))].copy())

    def optimize(self, solution=None):
        """optimizeCameraPoses with the recording ceres::Problem.  Returns the recorded problem, or None when the reference
        returned before building one.  `solution` = (q [M,4] wxyz memory order, t [M,3], X [P,3]) is installed in place of
        the Ceres solve."""
        if solution is None:
            rc = self.lib.ref_sys_optimize(self.h, None, None, None, 0, 0)
        else:
            q, t, X = (np.ascontiguousarray(a, np.float64) for a in solution)
            rc = self.lib.ref_sys_optimize(self.h, q.ctypes.data, t.ctypes.data, X.ctypes.data, len(q), len(X))
        if rc == 1:
            return None
        if rc != 0:
            raise RuntimeError("reference optimizeCameraPoses threw")
        info = np.zeros(5, np.int32)
        self.lib.ref_sys_problem_info(info)
        M, P, nr = int(info[0]), int(info[1]), int(info[2])
        q0, t0, X0, plane = np.zeros((M, 4)), np.zeros((M, 3)), np.zeros((P, 3)), np.zeros((P, 4))
        qc, tc, qt = np.zeros(M, np.int32), np.zeros(M, np.int32), np.zeros(M, np.int32)
        self.lib.ref_sys_problem_blocks(q0.reshape(-1), t0.reshape(-1), X0.reshape(-1), plane.reshape(-1), qc, tc, qt)
        kind, cam, point = np.zeros(nr, np.int32), np.zeros(nr, np.int32), np.zeros(nr, np.int32)
        r, loss = np.zeros((nr, 2)), np.zeros(nr)
        self.lib.ref_sys_problem_residuals(kind, cam, point, r.reshape(-1), loss)
        uv = np.zeros((nr, 2))
        self.lib.ref_sys_problem_uv(uv.reshape(-1))
        return dict(uv=uv, n_cams=M, n_points=P, max_iter=int(info[3]), linear_solver=int(info[4]), q0=q0, t0=t0, X0=X0, plane=plane,
                    q_const=qc, t_const=tc, q_tangent=qt, kind=kind, cam=cam, point=point, r=r, loss_a=loss,
                    cost0=float(self.lib.ref_sys_problem_cost()))
