"""Plain-numpy restatement of the reference's BA cost and the scipy minimiser used to pin the trust-region layer's
fixed point (tests/test_scipy_pin_extended.py, tests/golden/make_scipy_minima.py).  Shares no code with the product
or the oracle: camera models from the reference's headers (pinhole_camera_model.h:181-257,
pinhole_radial_tangential_camera_model.h:250-291, fisheye_camera_model.h:223-267), the functor from
reprojection_error.h:51-95 on scipy's Rotation, Ceres' loss functions from their documented definitions."""
import numpy as np
from scipy.optimize import least_squares
from scipy.optimize._numdiff import approx_derivative, group_columns
from scipy.sparse import lil_matrix
from scipy.spatial.transform import Rotation

from theiasfm_amd import abi, synth

NUM_EXT = 6


def _distort(model, K, q):
    """normalised camera-frame point -> pixel for one model; K [n, size], q [n, 3] (numpy, vectorised)"""
    f, ar, skew, px, py = K[:, 0], K[:, 1], K[:, 2], K[:, 3], K[:, 4]
    if model == abi.PINHOLE:  # pinhole_camera_model.h:181-210, 241-257
        n = q[:, :2] / q[:, 2:3]
        r2 = (n * n).sum(1)
        d = 1.0 + r2 * (K[:, 5] + K[:, 6] * r2)
        dx, dy = n[:, 0] * d, n[:, 1] * d
    elif model == abi.PINHOLE_RADIAL_TANGENTIAL:  # pinhole_radial_tangential_camera_model.h:250-291
        n = q[:, :2] / q[:, 2:3]
        x, y = n[:, 0], n[:, 1]
        r2 = x * x + y * y
        rd = 1.0 + K[:, 5] * r2 + K[:, 6] * r2 * r2 + K[:, 7] * r2 * r2 * r2
        t1, t2 = K[:, 8], K[:, 9]
        dx = x * rd + t2 * (r2 + 2.0 * x * x) + 2.0 * t1 * x * y
        dy = y * rd + t1 * (r2 + 2.0 * y * y) + 2.0 * t2 * x * y
    elif model == abi.FISHEYE:  # fisheye_camera_model.h:223-267 (takes the 3-D point)
        r = np.sqrt(q[:, 0] ** 2 + q[:, 1] ** 2)
        assert (r * r >= 1e-8).all()  # the pass-through branch is not exercised by these scenes
        th = np.arctan2(r, np.abs(q[:, 2]))
        t2 = th * th
        thd = th * (1.0 + K[:, 5] * t2 + K[:, 6] * t2 * t2 + K[:, 7] * t2 ** 3 + K[:, 8] * t2 ** 4)
        sgn = np.where(q[:, 2] < 0.0, -1.0, 1.0)
        dx, dy = sgn * thd * q[:, 0] / r, sgn * thd * q[:, 1] / r
    else:
        raise ValueError(model)
    return np.stack([f * dx + skew * dy + px, f * ar * dy + py], 1)


class Packed:
    """the free coordinates of a Problem as one vector, in scipy's terms"""

    def __init__(self, prob, point_dof):
        self.prob, self.dof = prob, point_dof
        nc, ng = prob.num_cameras, prob.num_groups
        self.free = [np.flatnonzero(prob.intrinsics_constant[prob.group_offset[g]:prob.group_offset[g + 1]] == 0)
                     for g in range(ng)]
        self.goff = np.concatenate([[0], np.cumsum([len(f) for f in self.free])]).astype(int)
        self.n_ext = NUM_EXT * nc
        self.n_intr = int(self.goff[-1])
        self.n = self.n_ext + self.n_intr + point_dof * prob.num_points

    def x0(self):
        p = self.prob
        intr = np.concatenate([p.intrinsics[p.group_offset[g] + f] for g, f in enumerate(self.free)]) if self.n_intr else np.zeros(0)
        return np.concatenate([p.extrinsics.ravel(), intr, p.points[:, :self.dof].ravel()])

    def blocks(self, x):
        """per-observation 2-vectors r_i = pixel - feature"""
        p = self.prob
        nc = p.num_cameras
        ext = x[:self.n_ext].reshape(nc, 6)
        intr = p.intrinsics.copy()
        for g, f in enumerate(self.free):
            intr[p.group_offset[g] + f] = x[self.n_ext + self.goff[g]:self.n_ext + self.goff[g + 1]]
        X = x[self.n_ext + self.n_intr:].reshape(p.num_points, self.dof)
        c, t = p.obs_camera, p.obs_point
        w = X[t, 3:4] if self.dof == 4 else 1.0
        q = Rotation.from_rotvec(ext[c, 3:]).apply(X[t, :3] - w * ext[c, :3])  # reprojection_error.h:69-81
        out = np.empty((len(c), 2))
        grp = p.camera_group[c]
        for model in np.unique(p.group_model):
            sel = np.flatnonzero(p.group_model[grp] == model)
            size = abi.INTRINSICS_SIZE[model]
            K = intr[p.group_offset[grp[sel]][:, None] + np.arange(size)[None, :]]
            out[sel] = _distort(model, K, q[sel])
        return out - p.obs_xy

    def sparsity(self, per_block_rows):
        p = self.prob
        m = len(p.obs_camera) * per_block_rows
        S = lil_matrix((m, self.n), dtype=int)
        for i, (c, t) in enumerate(zip(p.obs_camera, p.obs_point)):
            g = p.camera_group[c]
            cols = list(range(6 * c, 6 * c + 6)) + list(range(self.n_ext + self.goff[g], self.n_ext + self.goff[g + 1])) + \
                list(range(self.n_ext + self.n_intr + self.dof * t, self.n_ext + self.n_intr + self.dof * (t + 1)))
            for r in range(per_block_rows):
                S[per_block_rows * i + r, cols] = 1
        return S


SCIPY_LOSS = {abi.LOSS_HUBER: "huber", abi.LOSS_SOFTLONE: "soft_l1", abi.LOSS_CAUCHY: "cauchy", abi.LOSS_ARCTAN: "arctan"}


def rho(loss, s, a):
    """Ceres' loss functions of the squared block norm s (ceres/loss_function.h as documented; scale a)"""
    if loss == abi.LOSS_TRIVIAL:
        return s
    if loss == abi.LOSS_HUBER:
        return np.where(s <= a * a, s, 2 * a * np.sqrt(s) - a * a)
    if loss == abi.LOSS_SOFTLONE:
        return 2 * a * a * (np.sqrt(1 + s / (a * a)) - 1)
    if loss == abi.LOSS_CAUCHY:
        return a * a * np.log1p(s / (a * a))
    if loss == abi.LOSS_ARCTAN:
        return a * np.arctan2(s, a)
    raise ValueError(loss)


def scipy_minimum(prob, point_dof=3, loss=abi.LOSS_TRIVIAL, width=1.0, sparse=False):
    pk = Packed(prob, point_dof)
    kw, rows = {}, 2
    if loss == abi.LOSS_TRIVIAL:
        fun = lambda x: pk.blocks(x).ravel()  # noqa: E731
    else:
        def fun(x):
            r = pk.blocks(x)
            s = (r * r).sum(1)
            return (r * np.sqrt(rho(loss, s, width) / np.maximum(s, 1e-300))[:, None]).ravel()
    x0 = pk.x0()
    S = pk.sparsity(rows)
    if sparse:
        sol = least_squares(fun, x0, method="trf", jac_sparsity=S, tr_solver="lsmr", x_scale="jac",
                            xtol=1e-15, ftol=1e-13, gtol=1e-12, max_nfev=300, **kw)
        return sol.cost
    # finite differences on the known sparsity pattern (a few dozen evaluations per Jacobian instead of one per
    # variable), handed over dense so that scipy takes exact trust-region steps
    groups = group_columns(S)
    jac = lambda x: approx_derivative(fun, x, method="3-point", sparsity=(S, groups)).toarray()  # noqa: E731
    best = np.inf
    x = x0
    for _ in range(2):  # a restart from the last point: trf stops early on a flat gauge direction now and then
        sol = least_squares(fun, x, jac=jac, method="trf", x_scale="jac", xtol=1e-15, ftol=1e-15, gtol=1e-13, max_nfev=120, **kw)
        x = sol.x
        best = min(best, sol.cost)
    return best


def tight(point_dof, **kw):
    return abi.default_options(point_dof=point_dof, max_num_iterations=500, function_tolerance=1e-16,
                               gradient_tolerance=1e-14, parameter_tolerance=1e-14, use_inner_iterations=0, **kw)


def small(**kw):
    return synth.make_problem(6, 150, 720, seed=21, scene="ring", spread=1.0, **kw)


def case(name):
    """(problem, point_dof, loss, width, options) of a named case"""
    opt = {}
    dof, loss, width = 3, abi.LOSS_TRIVIAL, 1.0
    if name in ("huber", "softlone", "cauchy", "arctan"):
        prob = small()
        rng = np.random.default_rng(5)
        bad = rng.random(prob.num_observations) < 0.03
        prob.obs_xy[bad] += rng.normal(0.0, 12.0, (int(bad.sum()), 2))  # gross outliers for the loss to hold down
        loss = dict(huber=abi.LOSS_HUBER, softlone=abi.LOSS_SOFTLONE, cauchy=abi.LOSS_CAUCHY, arctan=abi.LOSS_ARCTAN)[name]
        width = 9.0 if name == "arctan" else 2.0
    elif name == "shared_group":
        prob = small(shared_group_size=3)  # two groups of three views, one intrinsics block each
        assert prob.num_groups == 2
    elif name == "fisheye_radtan_mix":
        prob = small(models=[(abi.FISHEYE, 0.5), (abi.PINHOLE_RADIAL_TANGENTIAL, 0.5)],
                     intrinsics_to_optimize=abi.INTRINSICS_FOCAL_LENGTH | abi.INTRINSICS_RADIAL_DISTORTION)
        assert set(prob.group_model.tolist()) == {abi.FISHEYE, abi.PINHOLE_RADIAL_TANGENTIAL}
    elif name == "shared_mixed_huber":
        prob = small(shared_group_size=2, models=[(abi.PINHOLE, 0.34), (abi.FISHEYE, 0.33), (abi.PINHOLE_RADIAL_TANGENTIAL, 0.33)],
                     intrinsics_to_optimize=abi.INTRINSICS_FOCAL_LENGTH | abi.INTRINSICS_PRINCIPAL_POINTS)
        loss, width = abi.LOSS_HUBER, 1.5
    elif name == "point_dof4":
        prob, dof = small(), 4
    elif name == "point_dof4_cauchy":
        prob, dof, loss, width = small(), 4, abi.LOSS_CAUCHY, 3.0
    else:
        raise ValueError(name)
    opt.update(loss_function_type=loss, robust_loss_width=width)
    return prob, dof, loss, width, opt


CASES = ["huber", "softlone", "cauchy", "arctan", "shared_group", "fisheye_radtan_mix", "shared_mixed_huber",
         "point_dof4", "point_dof4_cauchy"]


def ladybug():
    return synth.config("ladybug49")
