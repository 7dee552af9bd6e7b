"""The reference's OWN acceptance criterion for bundle adjustment, applied to this engine.

src/theia/sfm/incremental_reconstruction_estimator_test.cc:101-156 (and the global / hybrid estimator tests alike):
the bundle-adjusted fountain-11 reconstruction is aligned to data/sfm/gt_fountain11.bin with
AlignReconstructions (Umeyama on the camera positions of the common views, align_reconstructions.cc:97-130) and
every camera position must then lie within 1e-2 m of the ground truth.  The estimator pipeline that produces the
input of that final BA needs fountain11_matches.bin, which the reference does not ship; its OUTPUT
(data/sfm/fountain11.bin) is shipped.  So: start from the shipped reconstruction, push cameras and points away
from it by far more than the tolerance, run BundleAdjustReconstruction with the test's options
(intrinsics NONE; TRIVIAL and HUBER), align, and apply the reference's bound.

CPU: the oracle.  GPU: the device through the C ABI.  Golden: tests/golden/gt_fountain11_positions.json
(make_gt_fountain11_golden.py)."""
import json
import os

import numpy as np
import pytest

from theiasfm_amd import abi


def umeyama(src, dst):
    """Similarity (s, R, t) minimising sum |dst - (s R src + t)|^2 (AlignPointCloudsUmeyama,
    src/theia/sfm/transformation/align_point_clouds.cc; Umeyama 1991)."""
    mu_s, mu_d = src.mean(0), dst.mean(0)
    xs, xd = src - mu_s, dst - mu_d
    cov = xd.T @ xs / len(src)
    U, D, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    s = np.trace(np.diag(D) @ S) / (xs ** 2).sum(1).mean()
    return s, R, mu_d - s * R @ mu_s


def aligned_position_errors(prob, gt_pos):
    s, R, t = umeyama(prob.extrinsics[:, :3], gt_pos)
    return np.linalg.norm((s * (R @ prob.extrinsics[:, :3].T).T + t) - gt_pos, axis=1), s


def load(golden_dir):
    prob = abi.Problem.load(os.path.join(golden_dir, "fountain11_flat.npz"))
    gt = json.load(open(os.path.join(golden_dir, "gt_fountain11_positions.json")))
    return prob, np.array(gt["gt_positions_m"]), gt["tolerance_m"]


def perturbed(prob, gt_pos, seed=5):
    """Camera positions off by ~5 cm (five tolerances), orientations by ~0.2 degrees, points by ~5 cm."""
    _, scale = aligned_position_errors(prob, gt_pos)          # metres per reconstruction unit
    rng = np.random.default_rng(seed)
    p = prob.copy()
    p.extrinsics[:, :3] += rng.normal(0, 0.05 / scale, (p.num_cameras, 3))
    p.extrinsics[:, 3:] += rng.normal(0, 0.003, (p.num_cameras, 3))
    p.points[:, :3] += rng.normal(0, 0.05 / scale, (p.num_points, 3)) * p.points[:, 3:4]
    return p


CASES = [(abi.LOSS_TRIVIAL, abi.SPARSE_SCHUR), (abi.LOSS_HUBER, abi.SPARSE_SCHUR), (abi.LOSS_TRIVIAL, abi.ITERATIVE_SCHUR)]


def options(loss, solver, **kw):
    # ReconstructionEstimatorOptions of the reference test: intrinsics_to_optimize = NONE; BA defaults otherwise
    # (bundle_adjustment.h:78-122; SPARSE_SCHUR below 1000 views, reconstruction_estimator_utils.cc:121-130);
    # the incremental estimator switches inner iterations off (incremental_reconstruction_estimator.cc:516)
    return abi.default_options(loss_function_type=loss, robust_loss_width=2.0, linear_solver_type=solver,
                               point_dof=4, use_inner_iterations=0, **kw)


def test_shipped_reconstruction_meets_the_reference_bound(golden_dir):
    prob, gt_pos, tol = load(golden_dir)
    err, _ = aligned_position_errors(prob, gt_pos)
    assert err.max() < tol
    bad, _ = aligned_position_errors(perturbed(prob, gt_pos), gt_pos)
    assert bad.max() > 2 * tol      # the start of the BA below violates the bound: the BA has to earn it


@pytest.mark.parametrize("loss,solver", CASES)
def test_oracle_ba_meets_the_reference_bound(golden_dir, loss, solver):
    from oracle import oracle
    prob, gt_pos, tol = load(golden_dir)
    p = perturbed(prob, gt_pos)
    p.set_intrinsics_to_optimize(abi.INTRINSICS_NONE)
    st, s = oracle.solve(p, options(loss, solver))
    assert st == 0 and s.success == 1
    err, _ = aligned_position_errors(p, gt_pos)
    assert err.max() < tol, err
    assert 0.43 < s.final_rmse < 0.45      # the shipped reconstruction: 0.442277 (a robust loss trades a little RMSE)


@pytest.mark.gpu
@pytest.mark.parametrize("loss,solver", CASES)
def test_device_ba_meets_the_reference_bound(golden_dir, loss, solver):
    from oracle import oracle
    from theiasfm_amd import lib
    prob, gt_pos, tol = load(golden_dir)
    p = perturbed(prob, gt_pos)
    p.set_intrinsics_to_optimize(abi.INTRINSICS_NONE)
    q = p.copy()
    st, s = lib.solve(p, options(loss, solver, device=0))
    assert st == 0 and s.success == 1, s.message
    err, _ = aligned_position_errors(p, gt_pos)
    assert err.max() < tol, err
    assert 0.43 < s.final_rmse < 0.45
    # and the oracle from the same start ends at the same place
    st2, s2 = oracle.solve(q, options(loss, solver))
    assert st2 == 0
    assert abs(s.final_cost - s2.final_cost) < 1e-8 * s2.final_cost
    assert np.abs(aligned_position_errors(q, gt_pos)[0] - err).max() < 1e-6
