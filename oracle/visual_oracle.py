"""CPU oracle for the visual bundle-adjustment stage.  TEST INFRASTRUCTURE ONLY (same rules as balm_oracle.py).

Restates the reference's visual problem (paths relative to /root/reference):
  ReprojErrorWhitenedDistorted::operator()   include/utils.hpp:61-111  (Brown-Conrady, whitened by sigma_px)
  PointPlaneErrorWhitened::operator()        include/utils.hpp:133-139 (sqrt(s^2 + 1e-12) / sigma_plane)
  problem construction                       src/lvba_system.cpp:1571-1640 (camera 0 constant, landmarks without a
                                             valid plane dropped WITH their reprojection residuals, no loss function)
  solver options                             src/lvba_system.cpp:1572-1576 (DENSE_SCHUR, 50 iterations, Ceres defaults)
  write-back                                 src/lvba_system.cpp:1651-1665

THE TWO COST FUNCTORS ARE PINNED against the reference's own include/utils.hpp, compiled from /root/reference with the
stand-ins of oracle/shim and differentiated with forward-mode Jets as ceres::AutoDiffCostFunction would
(tests/test_ref_pin.py: residuals and ambient Jacobians to 1e-11, also against csrc/visual_math.h directly) -- except
ceres::QuaternionRotatePoint, which the stand-in restates from memory.  The problem construction (blocks, constancy,
manifold, residual set, sigmas, no robust loss) and its cost at the initial point are pinned against src/lvba_system.cpp:1571-1640 itself through a recording ceres::Problem (tests/test_ref_system.py).
THE SOLVER ITERATIONS ARE PARITY UNPINNED: its arithmetic lives in Ceres Solver 2.1.0 (README.md:20, find_package in
CMakeLists.txt:33), which is not in /root/reference and not installed here.  Its published algorithm is restated
FROM MEMORY of ceres-solver 2.1.0 (internal/ceres/trust_region_minimizer.cc, levenberg_marquardt_strategy.cc,
manifold.cc, rotation.h):
  * cost = 1/2 sum r^2; Jacobians by automatic differentiation (here: torch autograd, fp64) of the functors,
    chained with the manifold's PlusJacobian;
  * EigenQuaternionManifold: Plus(x, d) = q_d * x with q_d = [sin|d|/|d| * d, cos|d|] in Eigen (x,y,z,w) memory
    order; the reference stores [w,x,y,z] in that memory (src/lvba_system.cpp:1516 vs :1579) -- reproduced as is;
  * ceres::QuaternionRotatePoint normalises q before rotating (utils.hpp:72);
  * Jacobi scaling 1/(1+sqrt(colnorm^2)) fixed at iteration 0; LM diagonal sqrt(clamp(colnorm^2,1e-6,1e32)/radius);
    initial radius 1e4; accept if relative decrease > 1e-3; radius /= max(1/3, 1-(2 rho-1)^3) on success,
    radius /= decrease_factor (2,4,8,...) on failure AND on an invalid step (LevenbergMarquardtStrategy::StepIsInvalid is
    StepRejected(0); five invalid steps in a row are a FAILURE); parameter tolerance 1e-8, function tolerance 1e-6,
    checked in Ceres' order (parameter, function, then accept/reject); at the end of every iteration
    FinalizeIterationAndCheckIfMinimizerCanContinue tests max iterations, then gradient tolerance 1e-10, then min radius.
The solver restatement is pinned only by self-consistency tests (tests/test_visual_oracle.py): autograd Jacobians vs finite differences,
Schur-complement solve vs the full normal equations, manifold Plus/PlusJacobian consistency.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

F64 = torch.float64


# ----------------------------------------------------------------------------------------------- manifold
def eigen_quat_plus(a, d):
    """EigenQuaternionManifold::Plus on the 4-array `a` read as Eigen (x,y,z,w); d in R^3."""
    a = np.asarray(a, dtype=np.float64)
    d = np.asarray(d, dtype=np.float64)
    nd = np.linalg.norm(d)
    if nd == 0.0:
        return a.copy()
    s = np.sin(nd) / nd
    qd_v, qd_w = s * d, np.cos(nd)
    x_v, x_w = a[:3], a[3]
    out = np.empty(4)
    out[:3] = qd_w * x_v + x_w * qd_v + np.cross(qd_v, x_v)
    out[3] = qd_w * x_w - qd_v @ x_v
    return out


def eigen_quat_plus_jacobian(a):
    """EigenQuaternionManifold::PlusJacobian (4x3) at the 4-array a (memory order as given)."""
    x0, x1, x2, x3 = a
    return np.array([[x3, x2, -x1], [-x2, x3, x0], [x1, -x0, x3], [-x0, -x1, -x2]])


# ----------------------------------------------------------------------------------------------- residuals (torch)
def _rotate_wxyz(q, X):
    """ceres::QuaternionRotatePoint: normalise q = [w,x,y,z], then rotate."""
    q = q / q.norm()
    w, v = q[0], q[1:]
    uv = 2.0 * torch.linalg.cross(v, X)
    return X + w * uv + torch.linalg.cross(v, uv)


def reproj_residual(q, t, X, uv, intr, sigma):
    """utils.hpp:61-111.  intr = [fx, fy, cx, cy, k1, k2, p1, p2]."""
    fx, fy, cx, cy, k1, k2, p1, p2 = intr
    Xc = _rotate_wxyz(q, X) + t
    z = Xc[2]
    if float(z.detach()) <= 1e-8:
        return torch.zeros(2, dtype=F64) * (q.sum() + t.sum() + X.sum()) * 0.0
    xn, yn = Xc[0] / z, Xc[1] / z
    r2 = xn * xn + yn * yn
    r4 = r2 * r2
    radial = 1.0 + k1 * r2 + k2 * r4
    x_tan = 2.0 * p1 * xn * yn + p2 * (r2 + 2.0 * xn * xn)
    y_tan = p1 * (r2 + 2.0 * yn * yn) + 2.0 * p2 * xn * yn
    xd = xn * radial + x_tan
    yd = yn * radial + y_tan
    return torch.stack([(fx * xd + cx - uv[0]) / sigma, (fy * yd + cy - uv[1]) / sigma])


def plane_residual(X, plane, sigma):
    """utils.hpp:133-139; sigma already max(1e-9, sigma) (utils.hpp:131)."""
    r = 0.0 - (plane[0] * X[0] + plane[1] * X[1] + plane[2] * X[2] + plane[3])
    return torch.sqrt(r * r + 1e-12) / sigma


@dataclass
class VisualProblem:
    """Packed visual problem (same arrays as lvba_visual_create, include/lvba_hip.h)."""
    q: np.ndarray          # [M,4]  T_cam<-world rotation, memory order [w,x,y,z]
    t: np.ndarray          # [M,3]
    X: np.ndarray          # [T,3]  landmarks
    obs_off: np.ndarray    # [T+1]  CSR offsets of each landmark's reprojection observations
    obs_cam: np.ndarray    # [O]    camera of each observation
    obs_uv: np.ndarray     # [O,2]
    plane: np.ndarray      # [T,4]  (n, d)
    valid: np.ndarray      # [T]    landmark has a valid plane; others are dropped with their observations
    intr: np.ndarray       # [8]
    sigma_px: float = 0.5
    sigma_plane: float = 0.01

    def active(self):
        return np.nonzero(np.asarray(self.valid) != 0)[0]


class VisualOracle:
    """Dense Gauss-Newton/LM machinery for small problems: parameter vector in the TANGENT layout
    [cam 1..M-1: (dq 3, dt 3)] + [active landmarks: 3], camera 0 constant."""

    def __init__(self, p: VisualProblem):
        self.p = p
        self.M = p.q.shape[0]
        self.act = p.active()
        self.sig_pl = max(1e-9, p.sigma_plane)
        self.n_cam = 6 * (self.M - 1)
        self.n_par = self.n_cam + 3 * len(self.act)
        rows = []
        for li, ti in enumerate(self.act):
            for o in range(int(p.obs_off[ti]), int(p.obs_off[ti + 1])):
                rows.append(("r", li, int(ti), o))
            rows.append(("p", li, int(ti), -1))
        self.rows = rows

    # ambient state ---------------------------------------------------------------------------------------
    def state(self):
        return self.p.q.copy(), self.p.t.copy(), self.p.X.copy()

    def residuals_and_jacobian(self, q, t, X, want_jac=True):
        p = self.p
        intr = [float(v) for v in p.intr]
        res, Jrows = [], []
        for kind, li, ti, o in self.rows:
            Xt = torch.tensor(X[ti], dtype=F64, requires_grad=want_jac)
            if kind == "r":
                c = int(p.obs_cam[o])
                qt = torch.tensor(q[c], dtype=F64, requires_grad=want_jac)
                tt = torch.tensor(t[c], dtype=F64, requires_grad=want_jac)
                r = reproj_residual(qt, tt, Xt, torch.tensor(p.obs_uv[o], dtype=F64), intr, p.sigma_px)
                if want_jac:
                    J = np.zeros((2, self.n_par))
                    for k in range(2):
                        gq, gt, gX = torch.autograd.grad(r[k], (qt, tt, Xt), retain_graph=True, allow_unused=True)
                        gq = np.zeros(4) if gq is None else gq.numpy()
                        gt = np.zeros(3) if gt is None else gt.numpy()
                        gX = np.zeros(3) if gX is None else gX.numpy()
                        if c > 0:
                            J[k, 6 * (c - 1):6 * (c - 1) + 3] = gq @ eigen_quat_plus_jacobian(q[c])
                            J[k, 6 * (c - 1) + 3:6 * (c - 1) + 6] = gt
                        J[k, self.n_cam + 3 * li:self.n_cam + 3 * li + 3] = gX
                    Jrows.append(J)
                res.append(r.detach().numpy())
            else:
                r = plane_residual(Xt, torch.tensor(p.plane[ti], dtype=F64), self.sig_pl)
                if want_jac:
                    (gX,) = torch.autograd.grad(r, (Xt,))
                    J = np.zeros((1, self.n_par))
                    J[0, self.n_cam + 3 * li:self.n_cam + 3 * li + 3] = gX.numpy()
                    Jrows.append(J)
                res.append(np.array([float(r)]))
        r = np.concatenate(res) if res else np.zeros(0)
        J = np.concatenate(Jrows, 0) if (want_jac and Jrows) else None
        return r, J

    def cost(self, q, t, X):
        r, _ = self.residuals_and_jacobian(q, t, X, want_jac=False)
        return 0.5 * float(r @ r)

    def plus(self, q, t, X, delta):
        q2, t2, X2 = q.copy(), t.copy(), X.copy()
        for c in range(1, self.M):
            d = delta[6 * (c - 1):6 * (c - 1) + 6]
            q2[c] = eigen_quat_plus(q[c], d[:3])
            t2[c] = t[c] + d[3:]
        for li, ti in enumerate(self.act):
            X2[ti] = X[ti] + delta[self.n_cam + 3 * li:self.n_cam + 3 * li + 3]
        return q2, t2, X2

    # linear algebra ------------------------------------------------------------------------------------------
    def solve_schur(self, J, r, D):
        """DENSE_SCHUR: eliminate the 3x3 landmark blocks of (J^T J + D^2) x = J^T r.  Returns x."""
        nc = self.n_cam
        A = J.T @ J + np.diag(D * D)
        g = J.T @ r
        B, E, C = A[:nc, :nc], A[:nc, nc:], A[nc:, nc:]
        Cinv = np.zeros_like(C)
        for i in range(len(self.act)):
            s = slice(3 * i, 3 * i + 3)
            Cinv[s, s] = np.linalg.inv(C[s, s])
        S = B - E @ Cinv @ E.T
        rhs = g[:nc] - E @ (Cinv @ g[nc:])
        xc = np.linalg.solve(S, rhs) if nc else np.zeros(0)
        xp = Cinv @ (g[nc:] - E.T @ xc)
        return np.concatenate([xc, xp])

    def gradient_max_norm(self, q, g_unscaled):
        """Ceres 2.1's gradient_max_norm: the max norm of the PROJECTED gradient step, || x - Plus(x, -g) ||_inf over the ambient
        parameters (trust_region_minimizer.cc) -- |g| itself on the Euclidean blocks, the four components of q - Plus(q, -g_rot) on
        a quaternion block.  g_unscaled: tangent gradient, cameras 1.. (6 each: rotation, translation), then the landmarks."""
        nc = self.n_cam
        gm = np.abs(g_unscaled[nc:]).max(initial=0.0)
        for c in range(1, self.M):
            gc = g_unscaled[6 * (c - 1):6 * c]
            gm = max(gm, np.abs(q[c] - eigen_quat_plus(q[c], -gc[:3])).max(), np.abs(gc[3:]).max())
        return float(gm)

    # Ceres 2.1 TrustRegionMinimizer + LevenbergMarquardtStrategy, as restated in the module docstring ---------
    def solve(self, max_iter=50, verbose=False):
        q, t, X = self.state()
        radius, decrease_factor = 1e4, 2.0
        min_diag, max_diag = 1e-6, 1e32
        r, J = self.residuals_and_jacobian(q, t, X)
        cost = 0.5 * float(r @ r)
        scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0)))           # jacobi scaling, fixed at iteration 0
        J = J * scale
        trace = [dict(iter=0, cost=cost, cost_change=0.0, step_norm=0.0, radius=radius, accepted=1, rho=0.0)]
        status = "NO_CONVERGENCE"
        g = J.T @ r
        # FinalizeIterationAndCheckIfMinimizerCanContinue checks iterations, then gradient, then radius
        if max_iter <= 0:
            return (q, t, X), trace, "NO_CONVERGENCE"
        if self.gradient_max_norm(q, g / scale) <= 1e-10:
            return (q, t, X), trace, "CONVERGENCE(gradient)"
        invalid_run = 0
        x_norm = float(np.sqrt((q[1:] ** 2).sum() + (t[1:] ** 2).sum() + (X[self.act] ** 2).sum()))
        it = 0
        while True:
            it += 1
            if it > max_iter:
                break
            diag = np.clip((J * J).sum(0), min_diag, max_diag)
            D = np.sqrt(diag / radius)
            x = self.solve_schur(J, r, D)
            step = -x
            if not np.all(np.isfinite(step)):
                # LevenbergMarquardtStrategy::StepIsInvalid = StepRejected(0): radius / 2, / 4, / 8 ... ; five in a row fail
                radius = radius / decrease_factor
                decrease_factor *= 2.0
                trace.append(dict(iter=it, cost=cost, cost_change=0.0, step_norm=0.0, radius=radius, accepted=0, rho=0.0))
                invalid_run += 1
                if invalid_run >= 5:
                    status = "FAILURE"
                    break
                if radius < 1e-32:
                    status = "CONVERGENCE(radius)"
                    break
                continue
            mr = J @ step
            model_cost_change = -float(mr @ (r + mr / 2.0))
            if model_cost_change <= 0.0:
                # LevenbergMarquardtStrategy::StepIsInvalid = StepRejected(0): radius / 2, / 4, / 8 ... ; five in a row fail
                radius = radius / decrease_factor
                decrease_factor *= 2.0
                trace.append(dict(iter=it, cost=cost, cost_change=0.0, step_norm=0.0, radius=radius, accepted=0, rho=0.0))
                invalid_run += 1
                if invalid_run >= 5:
                    status = "FAILURE"
                    break
                if radius < 1e-32:
                    status = "CONVERGENCE(radius)"
                    break
                continue
            invalid_run = 0
            delta = step * scale
            q2, t2, X2 = self.plus(q, t, X, delta)
            cand = self.cost(q2, t2, X2)
            step_norm = float(np.sqrt(((q2 - q) ** 2).sum() + ((t2 - t) ** 2).sum() + ((X2 - X) ** 2).sum()))
            if step_norm <= 1e-8 * (x_norm + 1e-8):
                trace.append(dict(iter=it, cost=cost, cost_change=cost - cand, step_norm=step_norm, radius=radius, accepted=0, rho=0.0))
                status = "CONVERGENCE(parameter)"
                break
            cost_change = cost - cand
            if abs(cost_change) <= 1e-6 * cost:
                trace.append(dict(iter=it, cost=cost, cost_change=cost_change, step_norm=step_norm, radius=radius, accepted=0, rho=0.0))
                status = "CONVERGENCE(function)"
                break
            rho = cost_change / model_cost_change
            if rho > 1e-3:
                q, t, X = q2, t2, X2
                x_norm = float(np.sqrt((q[1:] ** 2).sum() + (t[1:] ** 2).sum() + (X[self.act] ** 2).sum()))
                r, J = self.residuals_and_jacobian(q, t, X)
                J = J * scale
                cost = 0.5 * float(r @ r)
                radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
                decrease_factor = 2.0
                trace.append(dict(iter=it, cost=cost, cost_change=cost_change, step_norm=step_norm, radius=radius, accepted=1, rho=rho))
                g = J.T @ r
                if it >= max_iter:                      # MaxSolverIterationsReached comes before GradientToleranceReached
                    break
                if self.gradient_max_norm(q, g / scale) <= 1e-10:
                    status = "CONVERGENCE(gradient)"
                    break
            else:
                radius = radius / decrease_factor
                decrease_factor *= 2.0
                trace.append(dict(iter=it, cost=cand, cost_change=cost_change, step_norm=step_norm, radius=radius, accepted=0, rho=rho))
                if radius < 1e-32:
                    status = "CONVERGENCE(radius)"
                    break
            if verbose:
                print(trace[-1])
        # write-back (src/lvba_system.cpp:1651-1665): quaternions re-normalised, landmarks of valid tracks updated
        q = q / np.linalg.norm(q, axis=1, keepdims=True)
        return (q, t, X), trace, status
