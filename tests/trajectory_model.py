"""An independent model of the TRAJECTORY of the trust-region layer.

The scipy pins (test_scipy_pin*.py) hold where the Levenberg-Marquardt layer converges; nothing code-independent held
the path it takes: the radius update, the step-quality ratio, accepted / rejected sequences, the forcing sequence of
the inexact PCG solve.  Real Ceres cannot run in this image, so this file is the next best thing: a Levenberg-Marquardt
written from Ceres' DOCUMENTED rules (docs "Solving Non-linear Least Squares": trust-region loop, Levenberg-Marquardt
strategy, inexact steps / ITERATIVE_SCHUR, Jacobi scaling, loss-function corrector) on machinery that shares nothing
with `oracle/` or `theiasfm_amd/`:

  * residuals: plain numpy from the reference's formulas (reprojection_error.h:51-95, pinhole_camera_model.h:181-257,
    the first-order branch of ceres::AngleAxisRotatePoint included), written so that they also run on complex numbers;
  * Jacobian: COMPLEX-STEP differentiation (exact to round-off, no step-size error, no analytic derivative and no dual
    number in sight) -- 6 + #intrinsics + point_dof evaluations per Jacobian thanks to the block structure;
  * loss functions and their corrector from Ceres' documented definitions (Triggs' correction, corrector.cc);
  * the step: `exact` solves the FULL damped normal equations (J^T J + D^T D) d = -J^T r with SuperLU -- no Schur
    complement at all, which is what DENSE_SCHUR / SPARSE_SCHUR must equal to round-off; `pcg` forms the Schur
    complement as a DENSE matrix from scipy sparse products and runs the conjugate-gradient recurrences of Ceres'
    ConjugateGradientsSolver (Q-tolerance = eta, r-tolerance off, residual refreshed every 10 iterations) with the
    block-Jacobi preconditioner of the requested block shape -- the forcing sequence, restated independently.

tests/test_trajectory_pin.py holds oracle and device to this model per iteration: identical accepted / rejected /
invalid sequences, radii, and costs to 1e-9 relative.  It pins the restatement of Ceres' RULES; whether those rules are
Ceres 1.14's to the letter is what tools/make_ceres_golden.md remains for (parity unpinned at the Ceres boundary)."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from theiasfm_amd import abi

PINHOLE_SIZE = 7


def angle_axis_rotate(aa, pt):
    """ceres::AngleAxisRotatePoint (rotation.h) on [n, 3] arrays, real or complex"""
    theta2 = (aa * aa).sum(1)
    big = theta2.real > np.finfo(float).eps
    out = np.empty_like(pt)
    if big.any():
        a, p = aa[big], pt[big]
        theta = np.sqrt(theta2[big])
        c, s = np.cos(theta)[:, None], np.sin(theta)[:, None]
        w = a / theta[:, None]
        out[big] = p * c + np.cross(w, p) * s + w * ((w * p).sum(1)[:, None] * (1.0 - c))
    if (~big).any():
        a, p = aa[~big], pt[~big]
        out[~big] = p + np.cross(a, p)
    return out


def loss_derivs(loss, a, s):
    """rho(s), rho'(s), rho''(s) of Ceres' loss functions (loss_function.h as documented; scale a)"""
    if loss == abi.LOSS_TRIVIAL:
        return s, np.ones_like(s), np.zeros_like(s)
    b = a * a
    if loss == abi.LOSS_HUBER:
        r = np.sqrt(np.maximum(s, 1e-300))
        out = s > b
        rho1 = np.where(out, np.maximum(np.finfo(float).tiny, a / r), 1.0)
        return np.where(out, 2.0 * a * r - b, s), rho1, np.where(out, -rho1 / (2.0 * np.maximum(s, 1e-300)), 0.0)
    if loss == abi.LOSS_SOFTLONE:
        t = 1.0 + s / b
        q = np.sqrt(t)
        return 2.0 * b * (q - 1.0), np.maximum(np.finfo(float).tiny, 1.0 / q), -1.0 / (2.0 * b * t * q)
    if loss == abi.LOSS_CAUCHY:
        t = 1.0 + s / b
        inv = 1.0 / t
        return b * np.log(t), np.maximum(np.finfo(float).tiny, inv), -(1.0 / b) * inv * inv
    raise ValueError(loss)


class Model:
    """PINHOLE problems without constant cameras / points (what the trajectory cases use); unknowns are ordered
    [extrinsics of every view | free intrinsics of every group | points]"""

    def __init__(self, prob, point_dof=3, loss=abi.LOSS_TRIVIAL, width=2.0):
        assert (prob.group_model == abi.PINHOLE).all() and not prob.camera_flags.any() and not prob.point_constant.any()
        self.p, self.dof, self.loss, self.width = prob, point_dof, loss, width
        nc, ng = prob.num_cameras, prob.num_groups
        self.free = [np.flatnonzero(prob.intrinsics_constant[prob.group_offset[g]:prob.group_offset[g + 1]] == 0) for g in range(ng)]
        self.goff = np.concatenate([[0], np.cumsum([len(f) for f in self.free])]).astype(int)
        self.n_ext, self.n_intr = 6 * nc, int(self.goff[-1])
        self.n_cam_side = self.n_ext + self.n_intr
        self.n = self.n_cam_side + point_dof * prob.num_points
        self.cam, self.pt = prob.obs_camera.astype(int), prob.obs_point.astype(int)
        self.grp = prob.camera_group[self.cam].astype(int)
        self.m = len(self.cam)

    def x0(self):
        p = self.p
        intr = np.concatenate([p.intrinsics[p.group_offset[g] + f] for g, f in enumerate(self.free)]) if self.n_intr else np.zeros(0)
        return np.concatenate([p.extrinsics.ravel(), intr, p.points[:, :self.dof].ravel()])

    def blocks(self, x):
        """r_i = projection - feature per observation, [m, 2]; works on complex x"""
        p = self.p
        ext = x[:self.n_ext].reshape(-1, 6)
        intr = p.intrinsics.astype(x.dtype)
        for g, f in enumerate(self.free):
            intr[p.group_offset[g] + f] = x[self.n_ext + self.goff[g]:self.n_ext + self.goff[g + 1]]
        X = x[self.n_cam_side:].reshape(-1, self.dof)
        w = X[self.pt, 3:4] if self.dof == 4 else 1.0
        q = angle_axis_rotate(ext[self.cam, 3:], X[self.pt, :3] - w * ext[self.cam, :3])  # reprojection_error.h:69-81
        K = intr[p.group_offset[self.grp][:, None] + np.arange(PINHOLE_SIZE)[None, :]]
        n = q[:, :2] / q[:, 2:3]                                                           # pinhole_camera_model.h:181-210
        r2 = (n * n).sum(1)
        d = 1.0 + r2 * (K[:, 5] + K[:, 6] * r2)                                            # :241-257
        dx, dy = n[:, 0] * d, n[:, 1] * d
        px = K[:, 0] * dx + K[:, 2] * dy + K[:, 3]
        py = K[:, 0] * K[:, 1] * dy + K[:, 4]
        return np.stack([px, py], 1) - p.obs_xy

    def jacobian(self, x):
        """d r / d x by complex steps, CSR [2 m, n]; one evaluation per column CLASS (every observation depends on one
        view, one group, one point, so the a-th coordinate of all views can be stepped at once)"""
        h = 1e-30
        rows, cols, vals = [], [], []
        i2 = 2 * np.arange(self.m)

        def add(col_of_obs, sel, d):
            for k in range(2):
                rows.append(i2[sel] + k)
                cols.append(col_of_obs[sel])
                vals.append(d[sel, k])
        everything = np.arange(self.m)
        for a in range(6):
            xc = x.astype(complex)
            xc[a:self.n_ext:6] += 1j * h
            add(6 * self.cam + a, everything, self.blocks(xc).imag / h)
        nf = np.array([len(f) for f in self.free])
        for k in range(int(nf.max()) if len(nf) else 0):
            xc = x.astype(complex)
            has = np.flatnonzero(nf > k)
            xc[self.n_ext + self.goff[has] + k] += 1j * h
            add(self.n_ext + self.goff[self.grp] + k, np.flatnonzero(nf[self.grp] > k), self.blocks(xc).imag / h)
        for b in range(self.dof):
            xc = x.astype(complex)
            xc[self.n_cam_side + b::self.dof] += 1j * h
            add(self.n_cam_side + self.dof * self.pt + b, everything, self.blocks(xc).imag / h)
        return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(2 * self.m, self.n))

    def cost(self, x):
        r = self.blocks(x)
        return 0.5 * loss_derivs(self.loss, self.width, (r * r).sum(1))[0].sum()

    def linearize(self, x):
        """cost, corrected residual vector and corrected Jacobian (Ceres' Corrector, corrector.cc as documented:
        sqrt(rho') (J - alpha / s r r^T J), residual scaled by sqrt(rho') / (1 - alpha), alpha from
        0.5 alpha^2 - alpha - rho'' / rho' s = 0; alpha = 0 where rho'' <= 0)"""
        r = self.blocks(x)
        J = self.jacobian(x)
        s = (r * r).sum(1)
        rho, rho1, rho2 = loss_derivs(self.loss, self.width, s)
        if self.loss == abi.LOSS_TRIVIAL:
            return 0.5 * rho.sum(), r.ravel(), J
        sq = np.sqrt(rho1)
        curved = (s > 0.0) & (rho2 > 0.0)
        D = 1.0 + 2.0 * s * rho2 / rho1
        alpha = np.where(curved, 1.0 - np.sqrt(np.maximum(D, 0.0)), 0.0)
        res_scale = np.where(curved, sq / (1.0 - alpha), sq)
        asn = np.where(curved, alpha / np.where(s > 0.0, s, 1.0), 0.0)
        # per block: J <- sq (J - asn r (r^T J)): a [2 m, 2 m] block-diagonal matrix applied from the left
        i2 = 2 * np.arange(self.m)
        b00 = sq * (1.0 - asn * r[:, 0] * r[:, 0])
        b01 = sq * (-asn * r[:, 0] * r[:, 1])
        b11 = sq * (1.0 - asn * r[:, 1] * r[:, 1])
        B = sp.csr_matrix((np.concatenate([b00, b01, b01, b11]),
                           (np.concatenate([i2, i2, i2 + 1, i2 + 1]), np.concatenate([i2, i2 + 1, i2, i2 + 1]))),
                          shape=(2 * self.m, 2 * self.m))
        return 0.5 * rho.sum(), (r * res_scale[:, None]).ravel(), (B @ J).tocsr()

    # ---- the reduced (Schur) system, dense, for the inexact step ------------------------------------------
    def precond_blocks(self, shape):
        """index sets of the block-Jacobi preconditioner over the camera side.  "merged": one block per view =
        extrinsics + its PRIVATE intrinsics (the C ABI's TMI_BA_PRECOND_SCHUR_JACOBI), shared intrinsics blocks on
        their own; "parameter_blocks": Ceres' shape, every parameter block on its own"""
        p = self.p
        members = np.bincount(p.camera_group, minlength=p.num_groups)
        out = []
        for c in range(p.num_cameras):
            idx = list(range(6 * c, 6 * c + 6))
            g = p.camera_group[c]
            if shape == "merged" and members[g] == 1:
                idx += list(range(self.n_ext + self.goff[g], self.n_ext + self.goff[g + 1]))
            out.append(np.array(idx))
        for g in range(p.num_groups):
            if self.goff[g + 1] > self.goff[g] and (shape != "merged" or members[g] > 1):
                out.append(np.arange(self.n_ext + self.goff[g], self.n_ext + self.goff[g + 1]))
        return out


def conjugate_gradients(A, b, Minv_blocks, q_tolerance, max_iterations=500, reset_period=10, min_iterations=0):
    """Ceres' ConjugateGradientsSolver from x = 0 (as documented / conjugate_gradients_solver.cc): termination on
    zeta = k (Q_k - Q_{k-1}) / Q_k < q_tolerance, the r-tolerance test disabled (LevenbergMarquardtStrategy sets -1)"""
    n = len(b)
    x = np.zeros(n)
    r = b.copy()
    if np.linalg.norm(b) == 0.0:
        return x, 0, True

    def precond(v):
        z = np.empty_like(v)
        for idx, Mi in Minv_blocks:
            z[idx] = Mi @ v[idx]
        return z
    rho, Q0, p = 1.0, 0.0, None
    k = 0
    while True:
        k += 1
        z = precond(r)
        last_rho, rho = rho, r @ z
        if rho == 0.0 or not np.isfinite(rho):
            return x, k, False
        p = z.copy() if k == 1 else z + (rho / last_rho) * p
        q = A @ p
        pq = p @ q
        if not (pq > 0.0):
            return x, k, True  # "matrix is indefinite, no more progress": what has been accumulated stands
        alpha = rho / pq
        x = x + alpha * p
        r = b - A @ x if k % reset_period == 0 else r - alpha * q
        Q1 = -0.5 * (x @ (b + r))
        zeta = k * (Q1 - Q0) / Q1
        if zeta < q_tolerance and k >= min_iterations:
            return x, k, True
        Q0 = Q1
        if k >= max_iterations:
            return x, k, True


def levenberg_marquardt(model, *, solver="exact", precond="merged", max_num_iterations=50, function_tolerance=1e-6,
                        gradient_tolerance=1e-10, parameter_tolerance=1e-8, initial_radius=1e4, max_radius=1e12,
                        min_radius=1e-32, min_relative_decrease=1e-3, min_lm_diagonal=1e-6, max_lm_diagonal=1e32, eta=0.1,
                        max_consecutive_invalid=5):
    """Ceres' TrustRegionMinimizer + LevenbergMarquardtStrategy as documented.  Returns (rows, final x, why); rows as
    abi.TRACE_FIELDS"""
    x = model.x0()
    cost, r, J = model.linearize(x)
    scale = 1.0 / (1.0 + np.sqrt(np.asarray(J.multiply(J).sum(0)).ravel()))   # Jacobi scaling, from the start point, kept
    S = sp.diags(scale)
    J = (J @ S).tocsr()
    g = J.T @ r
    rows = []
    if np.abs(g / scale).max() <= gradient_tolerance:
        return rows, x, "gradient tolerance reached"
    radius, decrease_factor, invalid_run = initial_radius, 2.0, 0
    nc = model.n_cam_side
    why = "maximum number of iterations reached"
    it = 0
    while it < max_num_iterations:
        it += 1
        H = (J.T @ J).tocsc()
        diag = H.diagonal()
        D2 = np.clip(diag, min_lm_diagonal, max_lm_diagonal) / radius
        Hd = (H + sp.diags(D2)).tocsc()
        lin_its, ok = 0, True
        if solver == "exact":
            try:
                delta = spla.splu(Hd).solve(-g)
                ok = bool(np.isfinite(delta).all())
            except RuntimeError:
                ok, delta = False, np.zeros_like(g)
        else:
            # eliminate the points: S = B - E C^-1 E^T (dense), b~ = -(g_c - E C^-1 g_p)
            B, E, Cb = Hd[:nc, :nc], Hd[:nc, nc:], Hd[nc:, nc:]
            dof, npt = model.dof, model.p.num_points
            Cd = np.zeros((npt, dof, dof))
            Cc = Cb.tocoo()
            Cd[Cc.row // dof, Cc.row % dof, Cc.col % dof] = Cc.data
            Ci = np.linalg.inv(Cd)
            ii = (np.arange(npt)[:, None, None] * dof + np.arange(dof)[None, :, None]) + np.zeros((1, 1, dof), int)
            jj = (np.arange(npt)[:, None, None] * dof + np.arange(dof)[None, None, :]) + np.zeros((1, dof, 1), int)
            Cinv = sp.csr_matrix((Ci.ravel(), (ii.ravel(), jj.ravel())), shape=Cb.shape)
            ECi = (E @ Cinv).tocsr()
            Sd = (B - ECi @ E.T).toarray()
            rhs = -(g[:nc] - ECi @ g[nc:])
            blocks = []
            for idx in model.precond_blocks(precond):
                blocks.append((idx, np.linalg.inv(Sd[np.ix_(idx, idx)])))
            dc, lin_its, ok = conjugate_gradients(Sd, rhs, blocks, eta)
            dp = Cinv @ (-g[nc:] - E.T @ dc)
            delta = np.concatenate([dc, dp])
        Jd = J @ delta
        mcc = -(Jd @ (r + 0.5 * Jd)) if ok else 0.0
        if not ok or not (mcc > 0.0):
            rows.append([it, cost, radius, -1.0, np.nan, mcc, lin_its, 0.0])
            invalid_run += 1
            if invalid_run >= max_consecutive_invalid:
                why = "too many consecutive invalid steps"
                break
            radius /= decrease_factor
            decrease_factor *= 2.0
            if radius < min_radius:
                why = "minimum trust region radius reached"
                break
            continue
        invalid_run = 0
        step = delta * scale
        x_plus = x + step
        cand = model.cost(x_plus)
        step_norm, x_norm = np.linalg.norm(step), np.linalg.norm(x)
        if step_norm <= parameter_tolerance * (x_norm + parameter_tolerance):
            rows.append([it, cost, radius, 2.0, cand, mcc, lin_its, step_norm])
            why = "parameter tolerance reached"
            break
        if abs(cost - cand) <= function_tolerance * cost:
            rows.append([it, cost, radius, 3.0, cand, mcc, lin_its, step_norm])
            why = "function tolerance reached"
            break
        quality = (cost - cand) / mcc
        accepted = quality > min_relative_decrease
        rows.append([it, cost, radius, 1.0 if accepted else 0.0, cand, mcc, lin_its, step_norm])
        if accepted:
            x = x_plus
            cost, r, J = model.linearize(x)
            J = (J @ S).tocsr()
            g = J.T @ r
            radius = min(max_radius, radius / max(1.0 / 3.0, 1.0 - (2.0 * quality - 1.0) ** 3))
            decrease_factor = 2.0
            if np.abs(g / scale).max() <= gradient_tolerance:
                why = "gradient tolerance reached"
                break
        else:
            radius /= decrease_factor
            decrease_factor *= 2.0
        if radius < min_radius:
            why = "minimum trust region radius reached"
            break
    return np.array(rows).reshape(-1, abi.TRACE_STRIDE), x, why
