"""CPU oracle of the window-BA stage that turns raw scans + odometry into anchor frames.  TEST INFRASTRUCTURE ONLY.

Literal restatement of LvbaSystem::runWindowBA (reference src/lvba_system.cpp:204-310): per window of `window_size`
frames -- cut_voxel / recut / tras_opt at the odometry poses (:247-257), skip if fewer than 3 voxels per frame (:258-262),
BALM2::damping_iter (:264), re-alignment of the optimised window to the odometry pose of its first frame (:268-279),
relative poses to the anchor (:284-299), merge of the transformed clouds with fp32 write-back (pl_transform,
include/BALM/tools.hpp:385-395) and down_sampling_voxel2 (tools.hpp:300-359).
down_sampling_voxel2 emits its survivors in unordered_map order (unspecified); here: sorted by voxel key (x, y, z).
PINNED: the pieces (map build, damping_iter, pl_transform, down_sampling_voxel2 with bit-identical survivors) against the
reference's BALM headers (tests/test_ref_pin.py), and the loops themselves -- runWindowBA, runLidarBA, the anchor merge of
optimizeCameraPoses -- against src/lvba_system.cpp compiled as it lies (oracle/ref_glue_system.cpp, tests/test_ref_system.py:
refined poses to 1e-12, anchor indices and relative poses equal).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def down_sampling_voxel2(pts, voxel_size):
    """tools.hpp:300-359 on [n,3] fp32 points: per voxel keep the ORIGINAL point closest to the voxel centre (first one on
    ties).  Returns the kept rows' indices, ordered by voxel key."""
    pts = np.asarray(pts, f32).reshape(-1, 3)
    if voxel_size < 0.001 or len(pts) == 0:
        return np.arange(len(pts))
    loc = (pts.astype(np.float64) / voxel_size).astype(f32)
    loc = np.where(loc < 0, (loc - f32(1.0)).astype(f32), loc)
    key = np.trunc(loc).astype(np.int64)
    c = (key.astype(np.float64) + 0.5) * voxel_size
    d = pts.astype(np.float64) - c
    d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
    order = np.lexsort((np.arange(len(pts)), d2, key[:, 2], key[:, 1], key[:, 0]))
    k = key[order]
    first = np.ones(len(order), bool)
    first[1:] = np.any(k[1:] != k[:-1], axis=1)
    return order[first]


def run_window_ba(clouds, poses, window_size, voxel_size, eigen_ratio, anchor_leaf, use_rel=True):
    """Returns dict(anchor_poses [A,12], anchor_clouds [A] of [m,3] fp32, anchor_index [n], rel_poses [n,12],
    window_poses [n,12] (optimised, before alignment; odometry for skipped windows), windows = per-window dicts)."""
    import oracle
    n = len(clouds)
    poses = np.asarray(poses, np.float64).reshape(n, 12)
    I12 = np.concatenate([np.eye(3).reshape(-1), np.zeros(3)])
    rel_poses = np.tile(I12, (n, 1))
    anchor_index = -np.ones(n, np.int32)
    window_poses = poses.copy()
    anchor_poses, anchor_clouds, windows = [], [], []
    for start in range(0, n, window_size):
        end = min(start + window_size, n)
        cw = end - start
        x_odom = poses[start:end]
        vm = oracle.voxel_build_cpp([np.asarray(c, f32)[:, :3] for c in clouds[start:end]], x_odom, voxel_size, eigen_ratio)
        V = len(vm["off"]) - 1
        info = dict(start=start, n=cw, n_voxels=V, skipped=V < 3 * cw, trace=None)
        windows.append(info)
        if info["skipped"]:
            continue
        co = oracle.COracle(cw, vm["off"], vm["idx"], vm["clu"])
        x_opt, trace, rc = co.damping_iter(x_odom)
        info["trace"] = trace
        window_poses[start:end] = x_opt
        R = x_opt[:, :9].reshape(cw, 3, 3)
        p = x_opt[:, 9:]
        Ro = x_odom[:, :9].reshape(cw, 3, 3)
        po = x_odom[:, 9:]
        if use_rel:                                                       # :268-277
            R_align = Ro[0] @ R[0].T
            p_align = po[0] - R_align @ p[0]
            Ra = np.einsum("ij,njk->nik", R_align, R)
            pa = p @ R_align.T + p_align
        else:
            Ra, pa = Ro, po
        merged = []
        for j in range(cw):                                               # :284-299
            rel_R = Ro[0].T @ Ra[j]
            rel_p = Ro[0].T @ (pa[j] - po[0])
            rel_poses[start + j] = np.concatenate([rel_R.reshape(-1), rel_p])
            anchor_index[start + j] = len(anchor_poses)
            c = np.asarray(clouds[start + j], f32)[:, :3].astype(np.float64)
            merged.append((c @ rel_R.T + rel_p).astype(f32))              # pl_transform writes fp32 back
        merged = np.concatenate(merged)
        keep = down_sampling_voxel2(merged, anchor_leaf)
        anchor_poses.append(x_odom[0].copy())
        anchor_clouds.append(merged[keep])
    return dict(anchor_poses=np.asarray(anchor_poses).reshape(-1, 12), anchor_clouds=anchor_clouds,
                anchor_index=anchor_index, rel_poses=rel_poses, window_poses=window_poses, windows=windows)


def run_lidar_ba(clouds, poses, *, window_enable=True, window_size=10, anchor_leaf=0.1, use_rel=True, stage1_enable=True,
                 stage_voxel_size=(0.5, 0.5), stage_eigen_ratio=((0.3, 0.1, 0.06, 0.03), (0.08, 0.08, 0.08, 0.08)),
                 window_eigen_ratio=(0.3, 0.1, 0.06, 0.03)):
    """LvbaSystem::runLidarBA (src/lvba_system.cpp:312-410), compute only.  Returns (poses [n,12], report dict)."""
    import oracle
    n = len(clouds)
    poses = np.asarray(poses, np.float64).reshape(n, 12)
    if window_enable:
        w = run_window_ba(clouds, poses, window_size, stage_voxel_size[0], np.float32(window_eigen_ratio), anchor_leaf, use_rel)
        anchor_poses, anchor_clouds, aidx, rel = w["anchor_poses"].copy(), w["anchor_clouds"], w["anchor_index"], w["rel_poses"]
    else:                                                                  # :221-229
        I12 = np.concatenate([np.eye(3).reshape(-1), np.zeros(3)])
        anchor_poses, anchor_clouds = poses.copy(), [np.asarray(c, f32)[:, :3] for c in clouds]
        aidx, rel = np.arange(n, dtype=np.int32), np.tile(I12, (n, 1))
    report = dict(n_anchors=len(anchor_clouds), stages=[])
    for idx in range(0 if stage1_enable else 1, 2):                        # :356-389
        vm = oracle.voxel_build_cpp(anchor_clouds, anchor_poses, stage_voxel_size[idx], np.float32(stage_eigen_ratio[idx]))
        co = oracle.COracle(len(anchor_clouds), vm["off"], vm["idx"], vm["clu"])
        anchor_poses, trace, rc = co.damping_iter(anchor_poses)
        report["stages"].append(dict(stage=idx, n_voxels=len(vm["off"]) - 1, trace=trace))
    out = poses.copy()                                                     # :393-404
    for i in range(n):
        a = aidx[i]
        if a < 0:
            continue
        A, Lr = anchor_poses[a], rel[i]
        R = A[:9].reshape(3, 3) @ Lr[:9].reshape(3, 3)
        out[i] = np.concatenate([R.reshape(-1), A[:9].reshape(3, 3) @ Lr[9:] + A[9:]])
    return out, report


def merge_anchors(clouds, poses, window_size, anchor_leaf):
    """The anchor clouds of LvbaSystem::optimizeCameraPoses (src/lvba_system.cpp:1466-1489): per window of `window_size`
    scans, every scan moved into the frame of the window's first scan with the CURRENT poses (pl_transform: fp32 write-back),
    concatenated and thinned by down_sampling_voxel2.  Returns (anchor_poses [A,12], anchor_clouds [A] of [m,3] fp32; the
    survivors ordered by voxel key, where the reference emits them in unordered_map order)."""
    n = len(clouds)
    poses = np.asarray(poses, np.float64).reshape(n, 12)
    anchor_poses, anchor_clouds = [], []
    for start in range(0, n, window_size):
        end = min(start + window_size, n)
        R0, p0 = poses[start, :9].reshape(3, 3), poses[start, 9:]
        merged = []
        for j in range(start, end):
            rel_R = R0.T @ poses[j, :9].reshape(3, 3)
            rel_p = R0.T @ (poses[j, 9:] - p0)
            c = np.asarray(clouds[j], f32)[:, :3].astype(np.float64)
            merged.append((c @ rel_R.T + rel_p).astype(f32))
        merged = np.concatenate(merged)
        anchor_poses.append(poses[start].copy())
        anchor_clouds.append(merged[down_sampling_voxel2(merged, anchor_leaf)])
    return np.asarray(anchor_poses).reshape(-1, 12), anchor_clouds
