"""Deterministic synthetic LiDAR-BA problems in the packed format of include/lvba_hip.h.

Follows SURVEY.md §8(d): N poses on a closed 3-D Lissajous loop (~200 x 150 x 5 m, so world
coordinates reach the 100 m regime where lambda_min is a 1e8:1 cancellation), V plane voxels
each seen by k ~ 2+Poisson(3) (clipped to [2,16]) poses drawn from a band |i-home| <= W, 5 % of
voxels also seen from the far side of the loop (home + N/2) so the pose graph is banded plus
sparse far blocks.  Each observer contributes 15..120 points, generated in the world frame on
the plane (1 cm thickness noise), moved to the body frame with the ground-truth pose, ROUNDED
TO fp32 (the reference reads fp32 PCL points and widens them, bavoxel.hpp:806) and accumulated
in fp64 into (P = sum pp^T, v = sum p, n)  -- exactly what cut_voxel/PointCluster::push produce
(bavoxel.hpp:819-820, tools.hpp:428-433).

Deviation from the survey's sketch: the initial pose perturbation is N(0,0.02 deg)/N(0,1 cm)
instead of 0.5 deg / 5 cm, because at 30 m range 0.5 deg smears a 0.3-0.9 m plane patch by
26 cm and the merged voxel stops being a plane (the reference front-end would never admit it:
recut's lambda0/lambda2 test, bavoxel.hpp:351-352); the exact second-order Hessian is then
indefinite and every LM step is rejected.  At 0.02 deg / 1 cm (LiDAR-odometry quality) LM
converges in ~5 accepted iterations; 0.03 deg / 2 cm exercises the reject branch (tests).

torch is used only as an array library + RNG so the same code runs on CPU (tests, small) and
on the GPU (bench, 10M factors / ~700M points).  Outputs are numpy arrays.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def _exp_so3(w):
    th = w.norm(dim=-1, keepdim=True).clamp_min(1e-300)
    k = w / th
    K = torch.zeros(w.shape[:-1] + (3, 3), dtype=w.dtype, device=w.device)
    K[..., 0, 1], K[..., 0, 2] = -k[..., 2], k[..., 1]
    K[..., 1, 0], K[..., 1, 2] = k[..., 2], -k[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -k[..., 1], k[..., 0]
    th = th[..., None]
    I = torch.eye(3, dtype=w.dtype, device=w.device).expand_as(K)
    return I + torch.sin(th) * K + (1 - torch.cos(th)) * (K @ K)


def trajectory(n_poses, gen, device):
    """Ground-truth poses T_world<-body on a closed Lissajous loop."""
    f64 = torch.float64
    t = torch.arange(n_poses, dtype=f64, device=device) * (2 * math.pi / n_poses)
    p = torch.stack([100 * torch.sin(t), 75 * torch.sin(2 * t), 2.5 * torch.sin(3 * t)], -1)
    dx, dy = 100 * torch.cos(t), 150 * torch.cos(2 * t)
    yaw = torch.atan2(dy, dx)
    rp = torch.randn(n_poses, 2, generator=gen, dtype=f64, device=device) * math.radians(2.0)
    zeros = torch.zeros_like(yaw)
    Rz = _exp_so3(torch.stack([zeros, zeros, yaw], -1))
    Ry = _exp_so3(torch.stack([zeros, rp[:, 1], zeros], -1))
    Rx = _exp_so3(torch.stack([rp[:, 0], zeros, zeros], -1))
    return Rz @ Ry @ Rx, p


REVISITS = ("antipodal", "figure8", "chords", "lot")


def make_balm_problem(n_poses, n_voxels, *, seed=20250925, band=50, loop_frac=0.05,
                      k_extra_mean=3.0, k_max=16, npts=(15, 120), rot_sigma_deg=0.02,
                      trans_sigma=0.01, thickness=0.01, device="cpu", chunk=None, revisit="antipodal"):
    """Returns dict(poses_gt, poses_init [N,12], voxel_off [V+1] i64, pose_idx [F] i32,
    clusters [F,10] f64, plane_n [V,3], plane_c [V,3]).

    revisit: where the loop-closure voxels (loop_frac of them) find their far observers -- the SHAPE of the co-visibility graph
    beyond the ring of reach `band`:
      "antipodal" (the BASELINE configs): pose h sees what h + N/2 sees -- a regular pattern a band ordering folds;
      "figure8":  h <-> N - h (a road driven back in the other direction): mirror chords, also foldable;
      "chords":   six random places visited twice (segments of 2 band + 1 poses, a -> b): irregular chords;
      "lot":      eight short stretches (band / 2 poses) spread irregularly over the trajectory, ALL at the same place (a parking
                  lot crossed eight times): half of the voxels homed there are seen from other crossings -- a dense clique of
                  8 x band / 2 poses hanging on the ring, which no band ordering keeps narrow."""
    if revisit not in REVISITS:
        raise ValueError(f"revisit must be one of {REVISITS}")
    dev = torch.device(device)
    f64 = torch.float64
    g_struct = torch.Generator(device=dev).manual_seed(seed)
    g_noise = torch.Generator(device=dev).manual_seed(seed + 1)
    g_pose = torch.Generator(device=dev).manual_seed(seed + 2)
    N, V = int(n_poses), int(n_voxels)

    R_gt, p_gt = trajectory(N, g_struct, dev)

    # ---- structure: home pose, observer count, observer set ---------------------------
    W = min(band, (N - 1) // 2)
    band_sz = min(2 * W + 1, N)
    k_max = max(2, min(k_max, band_sz))
    home = torch.randint(0, N, (V,), generator=g_struct, device=dev).sort().values
    lam = torch.full((V,), float(k_extra_mean), dtype=f64, device=dev)
    k = (2 + torch.poisson(lam, generator=g_struct)).clamp(2, k_max).long()
    loops_ok = N >= 4 * W + 4
    is_loop = (torch.rand(V, generator=g_struct, device=dev) < loop_frac) & loops_ok

    def band_members(center, Wb=None):
        Wb = W if Wb is None else Wb
        bsz = min(2 * Wb + 1, N)
        c = center.clamp(Wb, N - 1 - Wb) if N > 2 * Wb else torch.full_like(center, Wb)
        r = torch.rand(V, bsz, generator=g_struct, device=dev)
        off = r.topk(min(k_max, bsz), dim=1, largest=False).indices  # distinct offsets
        if off.shape[1] < k_max:
            off = torch.cat([off, off[:, :1].expand(V, k_max - off.shape[1])], 1)   # (duplicates are dropped below)
        return (c[:, None] - Wb + off).clamp(0, N - 1)

    near = band_members(home)
    if revisit == "antipodal":
        far = band_members((home + N // 2) % N)
    elif revisit == "figure8":
        far = band_members((N - home) % N)
    else:
        rs = np.random.default_rng(seed + 7)
        if revisit == "chords":
            n_ch, seg = 6, 2 * W + 1
            a = np.sort(rs.choice(np.arange(seg, N - seg, seg), size=min(n_ch, max(1, (N - 2 * seg) // seg)), replace=False))
            b = rs.permutation(a)                                   # every visited place is some other place's second visit
            ta, tb = torch.as_tensor(a, device=dev), torch.as_tensor(b, device=dev)
            d = (home[:, None] - ta[None, :]).abs()
            near_a = d.min(1)
            on = (near_a.values <= W) & (tb[near_a.indices] != ta[near_a.indices])
            far = band_members(torch.where(on, tb[near_a.indices] + (home - ta[near_a.indices]), home))
            is_loop = on & (torch.rand(V, generator=g_struct, device=dev) < min(1.0, loop_frac * N / max(1, len(a) * seg))) & loops_ok
        else:   # "lot"
            n_lot, Ls = 8, max(4, W // 2)
            starts = np.sort(rs.choice(np.arange(Ls, N - 2 * Ls, 2 * Ls), size=min(n_lot, max(2, (N - 3 * Ls) // (2 * Ls))), replace=False))
            ts = torch.as_tensor(starts, device=dev)
            rel = home[:, None] - ts[None, :]
            inside = (rel >= 0) & (rel < Ls)
            on = inside.any(1)
            mine = inside.float().argmax(1)
            other = (mine + 1 + torch.randint(0, len(starts) - 1, (V,), generator=g_struct, device=dev)) % len(starts)
            far = band_members(torch.where(on, ts[other] + Ls // 2, home), Ls // 2)
            is_loop = on & (torch.rand(V, generator=g_struct, device=dev) < 0.5) & loops_ok
    n_far = torch.where(is_loop, (k // 2).clamp_min(1), torch.zeros_like(k))
    j = torch.arange(k_max, device=dev)[None, :]
    n_near = (k - n_far)[:, None]
    obs = torch.where(j < n_near, near, far.gather(1, (j - n_near).clamp(0, k_max - 1)))
    valid = j < k[:, None]
    obs = torch.where(valid, obs, torch.full_like(obs, N + 1)).sort(dim=1).values
    # drop accidental duplicates (only possible when near/far bands overlap)
    dup = torch.zeros_like(valid)
    dup[:, 1:] = obs[:, 1:] == obs[:, :-1]
    valid = (obs <= N - 1) & ~dup
    k = valid.sum(1)
    if revisit != "antipodal" and int(k.min()) < 2:
        # (a far observer that is also a near one -- revisited places closer than the band: the few voxels left with one observer go)
        keep = k >= 2
        home, obs, valid, k = home[keep], obs[keep], valid[keep], k[keep]
        V = int(keep.sum())
    assert int(k.min()) >= 2, "generator produced a voxel with < 2 observers"
    voxel_off = torch.zeros(V + 1, dtype=torch.int64, device=dev)
    voxel_off[1:] = k.cumsum(0)
    F = int(voxel_off[-1])
    pose_idx = obs[valid]                       # row-major -> ascending inside each voxel
    vox_of = torch.arange(V, device=dev)[:, None].expand(V, k_max)[valid]

    # ---- planes -------------------------------------------------------------------------
    center = p_gt[home] + (torch.rand(V, 3, generator=g_struct, dtype=f64, device=dev) - 0.5) * 60.0
    nrm = torch.randn(V, 3, generator=g_struct, dtype=f64, device=dev)
    nrm = nrm / nrm.norm(dim=1, keepdim=True)
    helper = torch.where((nrm[:, 0].abs() < 0.9)[:, None],
                         torch.tensor([1.0, 0, 0], dtype=f64, device=dev),
                         torch.tensor([0, 1.0, 0], dtype=f64, device=dev))
    e1 = torch.linalg.cross(nrm, helper)
    e1 = e1 / e1.norm(dim=1, keepdim=True)
    e2 = torch.linalg.cross(nrm, e1)
    ext = 0.3 + 0.6 * torch.rand(V, generator=g_struct, dtype=f64, device=dev)

    # ---- points -> clusters, chunked over factors -----------------------------------------
    n_lo, n_hi = npts
    n_pts = torch.randint(n_lo, n_hi + 1, (F,), generator=g_noise, device=dev)
    clusters = torch.empty(F, 10, dtype=f64, device=dev)
    if chunk is None:
        chunk = 1_000_000 if dev.type == "cuda" else 100_000
    jj = torch.arange(n_hi, device=dev)[None, :]
    for s in range(0, F, chunk):
        e = min(F, s + chunk)
        vo, pi, npt = vox_of[s:e], pose_idx[s:e], n_pts[s:e]
        m = e - s
        ab = (torch.rand(m, n_hi, 2, generator=g_noise, dtype=f64, device=dev) - 0.5) * ext[vo][:, None, None]
        zz = torch.randn(m, n_hi, generator=g_noise, dtype=f64, device=dev) * thickness
        pw = (center[vo][:, None, :] + ab[..., 0:1] * e1[vo][:, None, :]
              + ab[..., 1:2] * e2[vo][:, None, :] + zz[..., None] * nrm[vo][:, None, :])
        pb = torch.einsum('fji,fnj->fni', R_gt[pi], pw - p_gt[pi][:, None, :])   # R^T (pw - p)
        pb = pb.to(torch.float32).to(f64)                                          # PCL fp32 points
        pb = pb * (jj < npt[:, None])[..., None]
        c = clusters[s:e]
        c[:, 0] = (pb[..., 0] * pb[..., 0]).sum(1)
        c[:, 1] = (pb[..., 0] * pb[..., 1]).sum(1)
        c[:, 2] = (pb[..., 0] * pb[..., 2]).sum(1)
        c[:, 3] = (pb[..., 1] * pb[..., 1]).sum(1)
        c[:, 4] = (pb[..., 1] * pb[..., 2]).sum(1)
        c[:, 5] = (pb[..., 2] * pb[..., 2]).sum(1)
        c[:, 6:9] = pb.sum(1)
        c[:, 9] = npt.to(f64)
        del ab, zz, pw, pb

    # ---- initial poses ----------------------------------------------------------------------
    dth = torch.randn(N, 3, generator=g_pose, dtype=f64, device=dev) * math.radians(rot_sigma_deg)
    dp = torch.randn(N, 3, generator=g_pose, dtype=f64, device=dev) * trans_sigma
    R0 = R_gt @ _exp_so3(dth)
    p0 = p_gt + dp

    def pack(R, p):
        return torch.cat([R.reshape(N, 9), p], 1).cpu().numpy()

    return dict(
        n_poses=N,
        poses_gt=pack(R_gt, p_gt),
        poses_init=pack(R0, p0),
        voxel_off=voxel_off.cpu().numpy(),
        pose_idx=pose_idx.to(torch.int32).cpu().numpy(),
        clusters=clusters.cpu().numpy(),
        plane_n=nrm.cpu().numpy(),
        plane_c=center.cpu().numpy(),
    )


# BASELINE.json configs: name -> (N poses, V voxels); F ~= 5 V
CONFIGS = {
    "C2": (500, 400_000),
    "C3": (2_000, 2_000_000),
    "C4": (10_000, 10_000_000),
}


# =====================================================================================================
# Visual stage (SURVEY.md section 8(d) "Visual"): cameras ride on the same trajectory through the reference's
# extrinsics (config/config.yaml:15-20), intrinsics of config.yaml:2-12 at scale 0.5, tracks of 4 consecutive
# cameras, 0.5 px pixel noise, every landmark on a plane (the (n, d) prior of src/lvba_system.cpp:1531-1565).
# =====================================================================================================
REF_INTRINSICS = np.array([1293.56944 * 0.5, 1293.3155 * 0.5, 626.91359 * 0.5, 522.799224 * 0.5,
                           -0.076160, 0.123001, -0.00113, 0.000251])
REF_IMAGE_WH = (640, 512)
_RCL = np.array([[0.00610193, -0.999863, -0.0154172], [-0.00615449, 0.0153796, -0.999863],
                 [0.999962, 0.00619598, -0.0060598]])
_PCL = np.array([0.0194384, 0.104689, -0.0251952])
_TLI = np.array([0.04165, 0.02326, -0.0284])


def _rotmat_to_quat_wxyz(R):
    """Batched rotation matrix -> unit quaternion [w,x,y,z] (w >= 0), torch."""
    m00, m11, m22 = R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]
    w = torch.sqrt(torch.clamp(1 + m00 + m11 + m22, min=1e-30)) / 2
    x = torch.sqrt(torch.clamp(1 + m00 - m11 - m22, min=1e-30)) / 2
    y = torch.sqrt(torch.clamp(1 - m00 + m11 - m22, min=1e-30)) / 2
    z = torch.sqrt(torch.clamp(1 - m00 - m11 + m22, min=1e-30)) / 2
    x = torch.copysign(x, R[:, 2, 1] - R[:, 1, 2])
    y = torch.copysign(y, R[:, 0, 2] - R[:, 2, 0])
    z = torch.copysign(z, R[:, 1, 0] - R[:, 0, 1])
    q = torch.stack([w, x, y, z], 1)
    return q / q.norm(dim=1, keepdim=True)


def project_distorted(Xc, intr):
    """Pixel of camera-frame points, Brown-Conrady model of include/utils.hpp:86-105 (torch, batched)."""
    fx, fy, cx, cy, k1, k2, p1, p2 = [float(v) for v in intr]
    xn, yn = Xc[..., 0] / Xc[..., 2], Xc[..., 1] / Xc[..., 2]
    r2 = xn * xn + yn * yn
    radial = 1 + k1 * r2 + k2 * r2 * r2
    xd = xn * radial + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)
    yd = yn * radial + p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn
    return torch.stack([fx * xd + cx, fy * yd + cy], -1)


def make_visual_problem(n_cams, n_tracks, *, seed=20250925, track_len=4, pixel_sigma=0.5, rot_sigma_deg=0.05,
                        trans_sigma=0.02, point_sigma=0.05, invalid_frac=0.05, loop_poses=None, device="cpu"):
    """Returns dict(q [M,4] wxyz, t [M,3], X [T,3], obs_off [T+1] i64, obs_cam [O] i32, obs_uv [O,2], plane [T,4],
    valid [T] u8, intr [8], q_gt, t_gt, X_gt).  Camera 0 starts at its ground truth (it is held constant)."""
    dev = torch.device(device)
    f64 = torch.float64
    g = torch.Generator(device=dev).manual_seed(seed + 10)
    M, T = int(n_cams), int(n_tracks)
    # the cameras are the first M poses of a loop sampled with >= 2000 poses (~0.6 m apart), so that small test
    # problems still have consecutive cameras that share landmarks
    R_wi, p_wi = trajectory(max(M, 2000) if loop_poses is None else int(loop_poses), g, dev)
    R_wi, p_wi = R_wi[:M], p_wi[:M]
    Rcl = torch.tensor(_RCL, dtype=f64, device=dev)
    tci = Rcl @ torch.tensor(_TLI, dtype=f64, device=dev) + torch.tensor(_PCL, dtype=f64, device=dev)
    Rcw = Rcl @ R_wi.transpose(1, 2)                      # src/lvba_system.cpp:860-861
    tcw = -(Rcw @ p_wi[..., None])[..., 0] + tci
    W, H = REF_IMAGE_WH
    intr = REF_INTRINSICS
    L = max(1, min(track_len, M))
    # landmark seen from cameras c0 .. c0+L-1: back-project a pixel of the MIDDLE camera at a random depth
    c0 = torch.randint(0, M - L + 1, (T,), generator=g, device=dev)
    cm = c0 + L // 2
    depth = 3.0 + 22.0 * torch.rand(T, generator=g, dtype=f64, device=dev)
    u = (0.2 + 0.6 * torch.rand(T, generator=g, dtype=f64, device=dev)) * W
    v = (0.2 + 0.6 * torch.rand(T, generator=g, dtype=f64, device=dev)) * H
    Xc = torch.stack([(u - intr[2]) / intr[0] * depth, (v - intr[3]) / intr[1] * depth, depth], 1)
    X_gt = (Rcw[cm].transpose(1, 2) @ (Xc - tcw[cm])[..., None])[..., 0]
    cams = c0[:, None] + torch.arange(L, device=dev)[None, :]
    Xc_all = (Rcw[cams] @ X_gt[:, None, :, None])[..., 0] + tcw[cams]
    px = project_distorted(Xc_all, intr)
    vis = (Xc_all[..., 2] > 0.5) & (px[..., 0] > 0) & (px[..., 0] < W) & (px[..., 1] > 0) & (px[..., 1] < H)
    px = px + torch.randn(T, L, 2, generator=g, dtype=f64, device=dev) * pixel_sigma
    px = px.to(torch.float32).to(f64)                     # keypoints are float (include/utils.hpp:37-38)
    k = vis.sum(1)
    obs_off = torch.zeros(T + 1, dtype=torch.int64, device=dev)
    obs_off[1:] = k.cumsum(0)
    obs_cam = cams[vis].to(torch.int32)
    obs_uv = px[vis]
    n = torch.randn(T, 3, generator=g, dtype=f64, device=dev)
    n = n / n.norm(dim=1, keepdim=True)
    plane = torch.cat([n, -(n * X_gt).sum(1, keepdim=True)], 1)
    valid = (torch.rand(T, generator=g, device=dev) >= invalid_frac).to(torch.uint8)
    plane = plane * valid[:, None].to(f64)                # landmarks without a plane carry n = 0, d = 0 (:1549-1551)
    # initial values
    dth = torch.randn(M, 3, generator=g, dtype=f64, device=dev) * math.radians(rot_sigma_deg)
    dtr = torch.randn(M, 3, generator=g, dtype=f64, device=dev) * trans_sigma
    dth[0] = 0
    dtr[0] = 0
    R0 = _exp_so3(dth) @ Rcw
    t0 = tcw + dtr
    X0 = X_gt + torch.randn(T, 3, generator=g, dtype=f64, device=dev) * point_sigma
    return dict(q=_rotmat_to_quat_wxyz(R0).cpu().numpy(), t=t0.cpu().numpy(), X=X0.cpu().numpy(),
                obs_off=obs_off.cpu().numpy(), obs_cam=obs_cam.cpu().numpy(), obs_uv=obs_uv.cpu().numpy(),
                plane=plane.cpu().numpy(), valid=valid.cpu().numpy(), intr=intr.copy(),
                q_gt=_rotmat_to_quat_wxyz(Rcw).cpu().numpy(), t_gt=tcw.cpu().numpy(), X_gt=X_gt.cpu().numpy())


# ------------------------------------------------------------------------------------------------------------------
# Raw scans for the voxel front-end (lvba_voxmap_build): a box room with slanted panels and non-planar clutter.
# ------------------------------------------------------------------------------------------------------------------

def _rot_zyx(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    Ry = np.array([[cp, 0, sp], [0, 1.0, 0], [-sp, 0, cp]])
    Rx = np.array([[1.0, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx


def make_scans(n_frames, pts_per_frame, *, seed=20250925, room=(30.0, 20.0, 6.0), n_panels=6, n_blobs=12,
               clutter_frac=0.15, noise=0.01, rot_sigma_deg=0.02, trans_sigma=0.01, origin=(0.0, 0.0, 0.0),
               point_floats=3):
    """Synthetic window of LiDAR scans.  Returns dict(clouds=[n_i, point_floats] float32 body-frame arrays,
    poses=[N,12] (ground truth + odometry-grade noise), poses_gt=[N,12]).  The sensor flies a loop inside a
    room[0] x room[1] x room[2] box centred at `origin` (a non-zero origin exercises negative / large voxel keys);
    every ray hits the nearest of the 6 room faces and n_panels slanted rectangles; clutter_frac of the points are
    replaced by samples of Gaussian blobs (non-planar -> voxels that split or drop)."""
    rng = np.random.default_rng(seed)
    L = np.asarray(room, np.float64)
    org = np.asarray(origin, np.float64)
    # planes: (point c, unit normal n, in-plane axes a,b, half extents)
    quads = []
    for ax in range(3):
        for sgn in (-1.0, 1.0):
            n = np.zeros(3); n[ax] = sgn
            c = np.zeros(3); c[ax] = sgn * L[ax] / 2
            a = np.zeros(3); a[(ax + 1) % 3] = 1.0
            b = np.zeros(3); b[(ax + 2) % 3] = 1.0
            quads.append((c, n, a, b, L[(ax + 1) % 3] / 2, L[(ax + 2) % 3] / 2))
    for _ in range(n_panels):
        R = _rot_zyx(*rng.uniform(-np.pi, np.pi, 3))
        c = rng.uniform(-0.35, 0.35, 3) * L
        quads.append((c, R[:, 2], R[:, 0], R[:, 1], rng.uniform(0.8, 2.5), rng.uniform(0.8, 2.5)))
    blobs = rng.uniform(-0.4, 0.4, (max(n_blobs, 1), 3)) * L
    blob_sig = rng.uniform(0.15, 0.5, max(n_blobs, 1))

    t = np.arange(n_frames) * (2 * np.pi / max(n_frames, 1))
    pos = np.stack([0.3 * L[0] * np.cos(t), 0.3 * L[1] * np.sin(t), 0.1 * L[2] * np.sin(2 * t)], -1)
    clouds, poses, poses_gt = [], [], []
    for f in range(n_frames):
        R = _rot_zyx(t[f] + np.pi / 2, *np.radians(rng.normal(0, 2.0, 2)))
        p = pos[f]
        d = rng.normal(size=(pts_per_frame, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        best = np.full(pts_per_frame, np.inf)
        for c, n, a, b, ha, hb in quads:
            den = d @ n
            s = np.where(np.abs(den) > 1e-9, ((c - p) @ n) / np.where(den == 0, 1, den), np.inf)
            hit = p + s[:, None] * d
            ok = (s > 0.3) & (np.abs((hit - c) @ a) <= ha) & (np.abs((hit - c) @ b) <= hb)
            best = np.where(ok & (s < best), s, best)
        pw = p + best[:, None] * d + rng.normal(0, noise, (pts_per_frame, 3))
        nb = int(clutter_frac * pts_per_frame)
        if n_blobs > 0 and nb > 0:
            which = rng.integers(0, len(blobs), nb)
            pw[:nb] = blobs[which] + rng.normal(size=(nb, 3)) * blob_sig[which, None]
        pw = pw[np.isfinite(pw).all(1)]
        pw = pw[rng.permutation(len(pw))]
        body = ((pw - p) @ R).astype(np.float32)                   # R^T (pw - p)
        cloud = np.zeros((len(body), point_floats), np.float32)
        cloud[:, :3] = body
        if point_floats > 3:
            cloud[:, 3:] = rng.random((len(body), point_floats - 3), dtype=np.float32)   # payload the library must skip
        clouds.append(cloud)
        p_world = p + org
        poses_gt.append(np.concatenate([R.reshape(-1), p_world]))
        dR = _rot_zyx(*np.radians(rng.normal(0, rot_sigma_deg, 3)))
        poses.append(np.concatenate([(R @ dR).reshape(-1), p_world + rng.normal(0, trans_sigma, 3)]))
    return dict(clouds=clouds, poses=np.asarray(poses), poses_gt=np.asarray(poses_gt))
