"""An optimizer that shares no code with this repository as a pin of the LM layer's FIXED POINT.

The reference's numbers come out of Ceres, which cannot be run here: the
trust-region / Schur / PCG layer is compared with an in-repo restatement only.  What can be pinned
independently is where that layer converges: scipy's MINPACK Levenberg-Marquardt (`least_squares(
method="lm")`, finite-difference Jacobian) minimises the same cost -- 1/2 sum |pixel(camera, X) -
feature|^2 with the reference's pinhole model (pinhole_camera_model.h:181-210,
reprojection_error.h:51-95), written here in plain numpy on top of scipy's Rotation -- from the same
start.  The minimum VALUE is gauge invariant, so the converged costs must agree although the two
optimizers walk different paths and stop at different points of the gauge orbit."""
import numpy as np
import pytest
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation

from oracle import oracle
from theiasfm_amd import abi, synth

FREE_INTR = (0, 5, 6)  # focal length, k1, k2: OptimizeIntrinsicsType FOCAL_LENGTH | RADIAL_DISTORTION


def _problem():
    return synth.make_problem(6, 150, 720, seed=21, scene="ring", spread=1.0)


def _residuals(x, prob):
    nc, npt = prob.num_cameras, prob.num_points
    ext = x[:6 * nc].reshape(nc, 6)
    K = prob.intrinsics.reshape(nc, 7).copy()
    K[:, FREE_INTR] = x[6 * nc:9 * nc].reshape(nc, 3)
    X = x[9 * nc:].reshape(npt, 3)
    c, p = prob.obs_camera, prob.obs_point
    q = Rotation.from_rotvec(ext[c, 3:]).apply(X[p] - ext[c, :3])  # w = 1 (point_dof = 3)
    n = q[:, :2] / q[:, 2:3]
    r2 = (n * n).sum(1)
    d = 1.0 + K[c, 5] * r2 + K[c, 6] * r2 * r2
    dx, dy = n[:, 0] * d, n[:, 1] * d
    px = K[c, 0] * dx + K[c, 2] * dy + K[c, 3]
    py = K[c, 0] * K[c, 1] * dy + K[c, 4]
    return np.stack([px - prob.obs_xy[:, 0], py - prob.obs_xy[:, 1]], 1).ravel()


def _x0(prob):
    nc = prob.num_cameras
    return np.concatenate([prob.extrinsics.ravel(), prob.intrinsics.reshape(nc, 7)[:, FREE_INTR].ravel(),
                           prob.points[:, :3].ravel()])


def _scipy_minimum(prob):
    sol = least_squares(_residuals, _x0(prob), args=(prob,), method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15,
                        max_nfev=200000)
    # polish with the trust-region-reflective solver from there (different algorithm, same minimum)
    sol2 = least_squares(_residuals, sol.x, args=(prob,), method="trf", xtol=1e-15, ftol=1e-15, gtol=1e-14)
    return min(sol.cost, sol2.cost)


def _tight(**kw):
    return abi.default_options(point_dof=3, max_num_iterations=400, function_tolerance=1e-16, gradient_tolerance=1e-14,
                               parameter_tolerance=1e-14, use_inner_iterations=0, **kw)


def test_numpy_cost_function_is_the_oracles():
    prob = _problem()
    assert (prob.group_model == abi.PINHOLE).all() and prob.num_groups == prob.num_cameras
    r = _residuals(_x0(prob), prob)
    c, _, bad = oracle.cost(prob, abi.default_options(point_dof=3))
    assert bad == 0 and abs(0.5 * (r @ r) - c) <= 1e-9 * c


@pytest.mark.parametrize("solver", [abi.DENSE_SCHUR, abi.ITERATIVE_SCHUR])
def test_oracle_converges_to_the_minimum_scipy_finds(solver):
    prob = _problem()
    ref = _scipy_minimum(prob)
    st, s = oracle.solve(prob.copy(), _tight(linear_solver_type=solver))
    assert st == 0 and s.success == 1
    assert abs(s.final_cost - ref) <= 1e-8 * ref, (s.final_cost, ref, s.num_iterations)


@pytest.mark.gpu
@pytest.mark.parametrize("solver", [abi.DENSE_SCHUR, abi.ITERATIVE_SCHUR])
def test_device_converges_to_the_minimum_scipy_finds(solver):
    from theiasfm_amd import lib
    prob = _problem()
    ref = _scipy_minimum(prob)
    st, s = lib.solve(prob.copy(), _tight(linear_solver_type=solver))
    assert st == 0 and s.success == 1
    assert abs(s.final_cost - ref) <= 1e-8 * ref, (s.final_cost, ref, s.num_iterations)
