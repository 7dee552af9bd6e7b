"""Pins of the oracle to reference facts that do not need Ceres in the image (CPU).

(a) gradient at the fixture -- data/sfm/fountain11.bin is the OUTPUT of the reference's own
    pipeline, whose last step optimises every track with Ceres (estimate_track.cc:238-246 ->
    BundleAdjustTrack, bundle_adjustment.cc:96-107).  At a Ceres optimum of a track the gradient
    J_p^T r of the reference's residual vanishes, so evaluating OUR residual and point Jacobian
    there and finding ~0 pins both to what real Ceres minimised.  (The camera blocks are not at
    a tight optimum in the fixture: a full BA still gains 1 %, so they are not used here.)
(b) inner-iteration order -- bundle_adjuster.cc:193-200 hands Ceres the linear-solver
    ordering (groups 0 tracks / 1 intrinsics / 2 extrinsics, :346-371) REVERSED: extrinsics
    blocks, intrinsics blocks, points.  An independent Python emulation of one sweep in that
    order (block-by-block sub-problems through the oracle's LM) must reproduce
    oracle_inner_sweep, and the swapped order must not.
"""
import os

import numpy as np

from oracle import oracle
from theiasfm_amd import abi, synth


def point_gradient_stats(prob, r, Jp):
    """relative gradient |sum J_p^T r| / sum |J_p col| |r| per track coordinate."""
    pt = prob.obs_point
    g = np.zeros((prob.num_points, Jp.shape[2]))
    n = np.zeros_like(g)
    np.add.at(g, pt, np.einsum("nij,ni->nj", Jp, r))
    np.add.at(n, pt, np.sqrt((Jp ** 2).sum(1)) * np.linalg.norm(r, axis=1)[:, None])
    return np.abs(g) / np.maximum(n, 1e-300)


def test_fountain11_point_gradient_vanishes_on_oracle_jets(golden_dir):
    prob = abi.Problem.load(os.path.join(golden_dir, "fountain11_flat.npz"))
    r, J, valid = oracle.evaluate(prob)
    assert valid.all()
    rel = point_gradient_stats(prob, r, J[:, :, 16:20])
    # measured: median 1.8e-9, 90 % of the coordinates below 1e-6 (a few tracks were added after
    # their last adjustment and are far from stationary -- they do not move the quantiles)
    assert np.median(rel) < 1e-7
    assert np.quantile(rel, 0.90) < 1e-5
    # the same statistic with a deliberately wrong Jacobian (sign of the w column, or the
    # pixel rows swapped) is orders of magnitude away: the pin discriminates
    bad = J[:, :, 16:20].copy()
    bad[:, :, :3] = bad[:, ::-1, :3]
    assert np.median(point_gradient_stats(prob, r, bad)) > 1e-3


def _inner_defaults(o):
    """Ceres' Minimizer::Options defaults the coordinate descent runs every block with."""
    return abi.default_options(
        loss_function_type=o.loss_function_type, robust_loss_width=o.robust_loss_width,
        linear_solver_type=abi.DENSE_QR, max_num_iterations=50, max_solver_time_in_seconds=1e9,
        use_inner_iterations=0, function_tolerance=1e-6, gradient_tolerance=1e-10,
        parameter_tolerance=1e-8, max_trust_region_radius=1e16, point_dof=o.point_dof)


def _solve_camera_side_block(P, o2, cams, free_extrinsics):
    """One extrinsics block (cams = [c]) or one intrinsics block (cams = views of the group),
    everything else constant; P is updated in place when the sub-solve is usable."""
    g = int(P.camera_group[cams[0]])
    a, b = int(P.group_offset[g]), int(P.group_offset[g + 1])
    sel = np.flatnonzero(np.isin(P.obs_camera, cams))
    if sel.size == 0:
        return
    order = np.argsort(np.searchsorted(np.asarray(cams), P.obs_camera[sel]), kind="stable")
    sel = sel[order]
    local_cam = np.searchsorted(np.asarray(cams), P.obs_camera[sel]).astype(np.int32)
    flags = np.array([P.camera_flags[c] if free_extrinsics else
                      (abi.CAMERA_POSITION_CONSTANT | abi.CAMERA_ORIENTATION_CONSTANT) for c in cams], np.uint8)
    iconst = np.ones(b - a, np.uint8) if free_extrinsics else P.intrinsics_constant[a:b].copy()
    Q = abi.Problem(P.extrinsics[cams].copy(), np.zeros(len(cams), np.int32), flags,
                    P.group_model[g:g + 1].copy(), np.array([0, b - a], np.int32),
                    P.intrinsics[a:b].copy(), iconst, P.points[P.obs_point[sel]].copy(),
                    np.ones(sel.size, np.uint8), local_cam, np.arange(sel.size, dtype=np.int32),
                    P.obs_xy[sel].copy())
    st, s = oracle.solve(Q, o2)
    if s.success:
        if free_extrinsics:
            P.extrinsics[cams] = Q.extrinsics
        else:
            P.intrinsics[a:b] = Q.intrinsics


def python_sweep(P, o, intrinsics_first=False):
    o2 = _inner_defaults(o)

    def extrinsics_set():
        for c in range(P.num_cameras):
            both = abi.CAMERA_POSITION_CONSTANT | abi.CAMERA_ORIENTATION_CONSTANT
            if (P.camera_flags[c] & both) != both:
                _solve_camera_side_block(P, o2, [c], True)

    def intrinsics_set():
        for g in range(P.num_groups):
            a, b = P.group_offset[g], P.group_offset[g + 1]
            cams = [int(c) for c in np.flatnonzero(P.camera_group == g)]
            if cams and not P.intrinsics_constant[a:b].all():
                _solve_camera_side_block(P, o2, cams, False)

    for f in ((intrinsics_set, extrinsics_set) if intrinsics_first else (extrinsics_set, intrinsics_set)):
        f()
    oracle.adjust_tracks(P, o2)


def _order_problem():
    prob = synth.make_problem(6, 120, 560, seed=5, scene="ring", spread=0.6, shared_group_size=2,
                              intrinsics_to_optimize=abi.INTRINSICS_FOCAL_LENGTH | abi.INTRINSICS_RADIAL_DISTORTION,
                              perturb=3.0)
    prob.intrinsics[prob.group_offset[:-1]] *= 1.02  # focal lengths off, so both camera sets have work
    return prob


def test_inner_sweep_follows_the_reversed_solver_ordering():
    prob = _order_problem()
    o = abi.default_options(point_dof=3)
    ref, swapped, emu, emu_swapped = prob.copy(), prob.copy(), prob.copy(), prob.copy()
    oracle.inner_sweep(ref, o)
    try:
        oracle.set_inner_order(1)
        oracle.inner_sweep(swapped, o)
    finally:
        oracle.set_inner_order(0)
    python_sweep(emu, o)
    python_sweep(emu_swapped, o, intrinsics_first=True)
    c = lambda p: oracle.cost(p, o)[0]  # noqa: E731
    c0 = c(prob)
    assert c(ref) < 0.5 * c0
    # the C sweep equals the independent emulation of the reference's order ...
    np.testing.assert_allclose(ref.extrinsics, emu.extrinsics, rtol=0, atol=1e-12)
    np.testing.assert_allclose(ref.intrinsics, emu.intrinsics, rtol=1e-13, atol=0)
    np.testing.assert_allclose(ref.points, emu.points, rtol=0, atol=1e-11)
    # ... the hook really swaps (same check on the other order) ...
    np.testing.assert_allclose(swapped.extrinsics, emu_swapped.extrinsics, rtol=0, atol=1e-12)
    # ... and the two orders are far apart on this problem, so the device test that compares
    # against this oracle can tell them apart
    assert np.abs(ref.extrinsics - swapped.extrinsics).max() > 1e-5
    assert abs(c(ref) - c(swapped)) > 1e-6 * c(ref)
