"""CPU oracle (numpy, fp64) for the BALM LiDAR bundle-adjustment hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  The product path is the HIP library behind
include/lvba_hip.h and must never route through this file.

PINNED AGAINST THE REFERENCE'S OWN CODE: the reference (xuankuzcr/Global-LVBA) ships no tests, golden vectors or
fixtures for this path, and Eigen / PCL are not installed here -- but its BALM headers
(include/BALM/tools.hpp, bavoxel.hpp) compile unmodified against the small Eigen / PCL stand-ins of oracle/shim
(`make -C oracle ref` -> oracle/_ref/libbalm_ref.so).  tests/test_ref_pin.py holds every function below against that
library (H, g to 1e-9; costs to 1e-9; LM-refined poses to 1e-5, the last LM decisions being taken at rounding-noise
level), and tests/golden/ref_balm.npz carries its answers to the GPU box.  What the stand-in supplies in Eigen's place
(3x3 symmetric eigen-solver, LDL^T, products) is the only part not the reference's own.  Additionally pinned by
(1) central finite differences of its own cost, (2) an independent C restatement (oracle/balm_oracle.c) and
(3) a structurally different autograd formulation (tests/test_oracle.py).

Each function cites the reference lines it restates (paths relative to /root/reference).

Packed problem format (shared with the C-ABI, include/lvba_hip.h):
  poses     [N][12] f64   R row-major (9) then p (3); T_world<-body        (tools.hpp:147-207)
  voxel_off [V+1]  i64    CSR offsets into the factor arrays
  pose_idx  [F]    i32    observing pose of each factor, ascending inside a voxel
  clusters  [F][10] f64   Pxx Pxy Pxz Pyy Pyz Pzz vx vy vz n  (PointCluster, tools.hpp:407-466)
A "factor" is one non-empty (voxel, pose) PointCluster slot of VOX_HESS::plvec_voxels.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

I3 = np.eye(3)


# ----------------------------------------------------------------------------- so(3) helpers
def hat(v):
    """tools.hpp:105-112."""
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def exp_so3(w):
    """Rodrigues, tools.hpp:62-77 (threshold 1e-11 -> identity)."""
    th = float(np.linalg.norm(w))
    if th >= 1e-11:
        K = hat(np.asarray(w, dtype=np.float64) / th)
        return I3 + math.sin(th) * K + (1.0 - math.cos(th)) * (K @ K)
    return I3.copy()


# ----------------------------------------------------------------------------- packing helpers
def unpack_clusters(clusters):
    """[F][10] -> P [F,3,3], v [F,3], n [F]."""
    c = np.asarray(clusters, dtype=np.float64).reshape(-1, 10)
    P = np.empty((c.shape[0], 3, 3))
    P[:, 0, 0] = c[:, 0]
    P[:, 0, 1] = P[:, 1, 0] = c[:, 1]
    P[:, 0, 2] = P[:, 2, 0] = c[:, 2]
    P[:, 1, 1] = c[:, 3]
    P[:, 1, 2] = P[:, 2, 1] = c[:, 4]
    P[:, 2, 2] = c[:, 5]
    return P, c[:, 6:9].copy(), c[:, 9].copy()


def pack_clusters(P, v, n):
    P = np.asarray(P)
    out = np.empty((P.shape[0], 10))
    out[:, 0] = P[:, 0, 0]
    out[:, 1] = P[:, 0, 1]
    out[:, 2] = P[:, 0, 2]
    out[:, 3] = P[:, 1, 1]
    out[:, 4] = P[:, 1, 2]
    out[:, 5] = P[:, 2, 2]
    out[:, 6:9] = v
    out[:, 9] = n
    return out


def unpack_poses(poses):
    x = np.asarray(poses, dtype=np.float64).reshape(-1, 12)
    return x[:, :9].reshape(-1, 3, 3).copy(), x[:, 9:12].copy()


def pack_poses(R, p):
    R = np.asarray(R)
    return np.concatenate([R.reshape(-1, 9), np.asarray(p).reshape(-1, 3)], axis=1)


def cluster_transform(P, v, n, R, p):
    """PointCluster::transform, tools.hpp:450-456."""
    v2 = R @ v + n * p
    rp = np.outer(R @ v, p)
    P2 = R @ P @ R.T + rp + rp.T + n * np.outer(p, p)
    return P2, v2, n


def push_voxel_admits(n_slots):
    """VOX_HESS::push_voxel, bavoxel.hpp:45-54: a voxel is admitted iff >= 2 slots are non-empty."""
    return int(np.count_nonzero(np.asarray(n_slots) != 0)) >= 2


@dataclass
class Problem:
    n_poses: int
    voxel_off: np.ndarray
    pose_idx: np.ndarray
    clusters: np.ndarray
    P: np.ndarray = field(init=False)
    v: np.ndarray = field(init=False)
    n: np.ndarray = field(init=False)

    def __post_init__(self):
        self.voxel_off = np.asarray(self.voxel_off, dtype=np.int64)
        self.pose_idx = np.asarray(self.pose_idx, dtype=np.int32)
        self.clusters = np.ascontiguousarray(self.clusters, dtype=np.float64).reshape(-1, 10)
        self.P, self.v, self.n = unpack_clusters(self.clusters)

    @property
    def n_voxels(self):
        return len(self.voxel_off) - 1


# ----------------------------------------------------------------------------- a4: value/grad/Hessian
def acc_evaluate2(prob: Problem, poses, head, end):
    """VOX_HESS::acc_evaluate2, bavoxel.hpp:68-174, for voxels [head, end).

    Returns (Hess [6N,6N], JacT [6N], residual).  Hess is the exact second-order Hessian
    of sum(lambda_min) (not J^T J); parameter order per pose [dtheta(3), dp(3)] with the
    right-multiplicative retraction R*Exp(dtheta), p+dp (bavoxel.hpp:725-726).
    """
    Rs, ps = unpack_poses(poses)
    N = prob.n_poses
    Hess = np.zeros((6 * N, 6 * N))
    JacT = np.zeros(6 * N)
    residual = 0.0
    for a in range(head, end):
        f0, f1 = int(prob.voxel_off[a]), int(prob.voxel_off[a + 1])
        fs = range(f0, f1)
        # :90-95  transform every non-empty slot and sum
        sigP = np.zeros((3, 3))
        sigv = np.zeros(3)
        sigN = 0.0
        for f in fs:
            i = prob.pose_idx[f]
            P2, v2, n2 = cluster_transform(prob.P[f], prob.v[f], prob.n[f], Rs[i], ps[i])
            sigP += P2
            sigv += v2
            sigN += n2
        # :97-103
        vBar = sigv / sigN
        lmbd, U = np.linalg.eigh(sigP / sigN - np.outer(vBar, vBar))
        NN = sigN
        u = [U[:, 0], U[:, 1], U[:, 2]]
        uk = u[0]
        ukukT = np.outer(uk, uk)
        # :107-110
        umumT = np.zeros((3, 3))
        for m in (1, 2):
            umumT += 2.0 / (lmbd[0] - lmbd[m]) * np.outer(u[m], u[m])
        Auk = {}
        viRiTuk = {}
        viRiTukukT = {}
        # :112-149
        for f in fs:
            i = int(prob.pose_idx[f])
            Pi, vi, ni, Ri = prob.P[f], prob.v[f], prob.n[f], Rs[i]
            vihat = hat(vi)
            RiTuk = Ri.T @ uk
            RiTukhat = hat(RiTuk)
            PiRiTuk = Pi @ RiTuk
            viRiTuk[f] = vihat @ RiTuk
            viRiTukukT[f] = np.outer(viRiTuk[f], uk)
            ti_v = ps[i] - vBar
            ukTti_v = float(uk @ ti_v)
            combo1 = hat(PiRiTuk) + vihat * ukTti_v
            combo2 = Ri @ vi + ni * ti_v
            A = np.zeros((3, 6))
            A[:, 0:3] = (Ri @ Pi + np.outer(ti_v, vi)) @ RiTukhat - Ri @ combo1
            A[:, 3:6] = np.outer(combo2, uk) + float(combo2 @ uk) * I3
            A /= NN
            Auk[f] = A
            jjt = A.T @ uk
            JacT[6 * i:6 * i + 6] += jjt
            HRt = 2.0 / NN * (1.0 - ni / NN) * viRiTukukT[f]
            Hb = A.T @ umumT @ A
            Hb[0:3, 0:3] += (2.0 / NN * (combo1 - RiTukhat @ Pi) @ RiTukhat
                             - 2.0 / NN / NN * np.outer(viRiTuk[f], viRiTuk[f])
                             - 0.5 * hat(jjt[0:3]))
            Hb[0:3, 3:6] += HRt
            Hb[3:6, 0:3] += HRt.T
            Hb[3:6, 3:6] += 2.0 / NN * (ni - ni * ni / NN) * ukukT
            Hess[6 * i:6 * i + 6, 6 * i:6 * i + 6] += Hb
        # :151-167  pose pairs i<j
        fl = list(fs)
        for x in range(len(fl) - 1):
            fi = fl[x]
            i = int(prob.pose_idx[fi])
            ni = prob.n[fi]
            for y in range(x + 1, len(fl)):
                fj = fl[y]
                j = int(prob.pose_idx[fj])
                nj = prob.n[fj]
                Hb = Auk[fi].T @ umumT @ Auk[fj]
                Hb[0:3, 0:3] += -2.0 / NN / NN * np.outer(viRiTuk[fi], viRiTuk[fj])
                Hb[0:3, 3:6] += -2.0 * nj / NN / NN * viRiTukukT[fi]
                Hb[3:6, 0:3] += -2.0 * ni / NN / NN * viRiTukukT[fj].T
                Hb[3:6, 3:6] += -2.0 * ni * nj / NN / NN * ukukT
                Hess[6 * i:6 * i + 6, 6 * j:6 * j + 6] += Hb
        residual += lmbd[0]  # :168
    # :171-173 mirror strictly-upper blocks to lower
    iu = np.triu_indices(N, 1)
    for i, j in zip(*iu):
        blk = Hess[6 * i:6 * i + 6, 6 * j:6 * j + 6]
        if blk.any():
            Hess[6 * j:6 * j + 6, 6 * i:6 * i + 6] = blk.T
    return Hess, JacT, residual


# ----------------------------------------------------------------------------- a5: cost only
def evaluate_only_residual(prob: Problem, poses):
    """VOX_HESS::evaluate_only_residual, bavoxel.hpp:176-203 (sum of lambda_min, not averaged)."""
    Rs, ps = unpack_poses(poses)
    residual = 0.0
    for a in range(prob.n_voxels):
        sigP = np.zeros((3, 3))
        sigv = np.zeros(3)
        sigN = 0.0
        for f in range(int(prob.voxel_off[a]), int(prob.voxel_off[a + 1])):
            i = prob.pose_idx[f]
            P2, v2, n2 = cluster_transform(prob.P[f], prob.v[f], prob.n[f], Rs[i], ps[i])
            sigP += P2
            sigv += v2
            sigN += n2
        vBar = sigv / sigN
        residual += np.linalg.eigvalsh(sigP / sigN - np.outer(vBar, vBar))[0]
    return residual


def voxel_lambdas(prob: Problem, poses):
    """Vectorised per-voxel eigenvalues (ascending) of the merged covariance; same math as
    evaluate_only_residual but O(F) numpy, for mid-size checks."""
    Rs, ps = unpack_poses(poses)
    idx = prob.pose_idx
    R, p = Rs[idx], ps[idx]
    Rv = np.einsum('fij,fj->fi', R, prob.v)
    v2 = Rv + prob.n[:, None] * p
    rp = np.einsum('fi,fj->fij', Rv, p)
    P2 = (np.einsum('fij,fjk,flk->fil', R, prob.P, R) + rp + rp.transpose(0, 2, 1)
          + prob.n[:, None, None] * np.einsum('fi,fj->fij', p, p))
    starts = prob.voxel_off[:-1]
    sP = np.add.reduceat(P2.reshape(-1, 9), starts, axis=0).reshape(-1, 3, 3)
    sv = np.add.reduceat(v2, starts, axis=0)
    sn = np.add.reduceat(prob.n, starts)
    vb = sv / sn[:, None]
    C = sP / sn[:, None, None] - np.einsum('vi,vj->vij', vb, vb)
    return np.linalg.eigvalsh(C)


# ----------------------------------------------------------------------------- a6/a7: BALM2 wrappers
THD_NUM = 16  # bavoxel.hpp:25


def thread_slices(g_size, thd_num=THD_NUM):
    """BALM2::divide_thread slicing, bavoxel.hpp:614-624 (1 slice if g_size < thd_num;
    double part, C truncation to int at the thread-argument conversion)."""
    t = thd_num if g_size >= thd_num else 1
    part = 1.0 * g_size / t
    return [(int(part * i), int(part * (i + 1))) for i in range(t)]


def divide_thread(prob: Problem, poses):
    """BALM2::divide_thread, bavoxel.hpp:597-639: per-slice H/g/residual summed in slice
    order; returns residual averaged over voxels (AVG_THR)."""
    N = prob.n_poses
    Hess = np.zeros((6 * N, 6 * N))
    JacT = np.zeros(6 * N)
    residual = 0.0
    for head, end in thread_slices(prob.n_voxels):
        H_t, g_t, r_t = acc_evaluate2(prob, poses, head, end)
        Hess += H_t
        JacT += g_t
        residual += r_t
    return Hess, JacT, residual / prob.n_voxels


def only_residual(prob: Problem, poses, is_avg=False):
    """BALM2::only_residual, bavoxel.hpp:641-648."""
    r = evaluate_only_residual(prob, poses)
    return r / prob.n_voxels if is_avg else r


# ----------------------------------------------------------------------------- a8: LM driver
@dataclass
class LMTraceRow:
    it: int
    residual1: float
    residual2: float
    u: float          # damping used for this solve
    v: float
    q: float          # residual1 - residual2
    q1: float         # predicted decrease (averaged)
    accepted: bool
    evaluated: bool   # H/g recomputed at the start of this iteration


def retract(poses, dx):
    """bavoxel.hpp:722-727."""
    Rs, ps = unpack_poses(poses)
    dx = np.asarray(dx).reshape(-1, 6)
    for j in range(Rs.shape[0]):
        Rs[j] = Rs[j] @ exp_so3(dx[j, 0:3])
        ps[j] = ps[j] + dx[j, 3:6]
    return pack_poses(Rs, ps)


def ldlt_solve(A, b):
    """Unpivoted dense LDL^T on the LOWER triangle (what Eigen::SimplicialLDLT factors,
    bavoxel.hpp:706-710, up to its fill-reducing permutation).  O(n^3) numpy; small n only."""
    A = np.array(A, dtype=np.float64)
    n = A.shape[0]
    L = np.tril(A, -1)
    d = np.zeros(n)
    for j in range(n):
        d[j] = A[j, j] - (L[j, :j] ** 2) @ d[:j]
        L[j + 1:, j] = (A[j + 1:, j] - (L[j + 1:, :j] * L[j, :j]) @ d[:j]) / d[j]
    L = L + np.eye(n)
    y = np.linalg.solve(L, b)
    return np.linalg.solve(L.T, y / d)


def damping_iter(prob: Problem, poses, max_iter=10, u0=0.01, v0=2.0, rel_tol=1e-6,
                 eval_fn=None, cost_fn=None, solve_fn=None):
    """BALM2::damping_iter, bavoxel.hpp:662-767.  Returns (poses, trace).

    eval_fn / cost_fn / solve_fn let tests swap in faster equivalents of divide_thread /
    only_residual / the LDLT solve while keeping the control flow under test.
    """
    eval_fn = eval_fn or (lambda x: divide_thread(prob, x))
    cost_fn = cost_fn or (lambda x: only_residual(prob, x, True))
    solve_fn = solve_fn or ldlt_solve
    x = np.array(poses, dtype=np.float64).reshape(-1, 12)
    V = prob.n_voxels
    u, v = u0, v0
    is_calc_hess = True
    trace = []
    Hess = JacT = None
    residual1 = 0.0
    for it in range(max_iter):
        evaluated = is_calc_hess
        if is_calc_hess:
            Hess, JacT, residual1 = eval_fn(x)           # :688-689
        D = np.diag(Hess).copy()                         # :692
        HessuD = Hess + u * np.diag(D)                   # :693
        dxi = solve_fn(HessuD, -JacT)                    # :695-710
        x_temp = retract(x, dxi)                         # :722-727
        q1 = 0.5 * float(dxi @ (u * D * dxi - JacT))     # :729
        residual2 = cost_fn(x_temp)                      # :731
        q1 /= V                                          # :732
        q = residual1 - residual2                        # :736
        row = LMTraceRow(it, residual1, residual2, u, v, q, q1, q > 0, evaluated)
        trace.append(row)
        if q > 0:                                        # :744-752
            x = x_temp
            qq = q / q1
            v = 2.0
            qq = 1.0 - (2.0 * qq - 1.0) ** 3
            u *= (1.0 / 3.0) if qq < (1.0 / 3.0) else qq
            is_calc_hess = True
        else:                                            # :753-758
            u = u * v
            v = 2.0 * v
            is_calc_hess = False
        if abs(residual1 - residual2) / residual1 < rel_tol:   # :760
            break
    return x, trace
