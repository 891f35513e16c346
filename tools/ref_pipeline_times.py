"""Times the stages of the REFERENCE'S OWN pipeline (src/lvba_system.cpp + src/dataset_io.cpp compiled against the stand-ins of
oracle/shim, see oracle/ref_glue_system.cpp) on the synthetic sequence of tests/test_gpu_pipeline.py -- the CPU numbers to put
beside the GPU pipeline's.  Needs /root/reference (or a prebuilt oracle/_ref/liblvba_system_ref.so).  Development helper: uses
oracle/, never imported by the product.

    python tools/ref_pipeline_times.py [n_frames pts_per_scan]"""
import importlib
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_gpu_pipeline as tp
    import test_ref_system as trs
    from oracle import ref_system as rs
    ds = importlib.import_module("global-lvba_amd.dataset")
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    pts = int(sys.argv[2]) if len(sys.argv) > 2 else 80000
    d = tp._dataset(n_frames=n_frames, pts=pts)
    root = os.path.join(tempfile.mkdtemp(), "seq")
    trs.write_sequence(root, d, ds)
    trs.INTR, trs.W, trs.H = tp.INTR, tp.W, tp.H
    out = dict(n_scans=n_frames, points_per_scan=pts, n_images=n_frames, cores=1,
               note="reference sources, stand-in Eigen/PCL/OpenCV (no vectorised Eigen kernels), -O2, single thread except BALM's 16 std::threads")

    def timed(name, fn):
        t0 = time.perf_counter()
        r = fn()
        out[name + "_s"] = round(time.perf_counter() - t0, 4)
        return r
    S = timed("dataset_io", lambda: rs.ReferenceSystem(root, trs.reference_params(tp)))
    S.init()
    timed("run_lidar_ba", S.run_lidar_ba)
    timed("build_grid_map", S.build_grid_map)
    S.update_camera_poses()
    timed("generate_depth", lambda: S.generate_depth(tp.W, tp.H))
    S.set_features(d["kps"], {pr: m for pr, m in zip(d["pairs"], d["matches"])})
    tracks = timed("build_tracks_and_fuse", S.build_tracks)
    P = timed("optimize_camera_poses_up_to_solve", S.optimize)
    out.update(n_tracks=len(tracks), n_landmarks_with_plane=P["n_points"], n_residual_blocks=len(P["kind"]))
    S.close()
    shutil.rmtree(os.path.dirname(root), ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
