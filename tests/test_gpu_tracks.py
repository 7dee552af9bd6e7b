"""-m gpu: the per-track side kernels (post-BA outlier filter, batched BundleAdjustTrack)
through the C ABI against the CPU oracle.

Tolerances: flags / termination codes / iteration counts are integers and must be equal;
mean squared reprojection errors and costs 1e-9 relative (fp64 both sides, the device fuses
multiply-adds); adjusted points 1e-8 relative to the scene scale (BASELINE.json's bar for the
path is RMSE within 1e-6)."""
import numpy as np
import pytest

from oracle import oracle
from theiasfm_amd import abi, lib, synth

pytestmark = pytest.mark.gpu

MODELS = [(abi.PINHOLE, 0.4), (abi.PINHOLE_RADIAL_TANGENTIAL, 0.15), (abi.FISHEYE, 0.15),
          (abi.FOV, 0.15), (abi.DIVISION_UNDISTORTION, 0.15)]


def corrupted(seed, n_cam=24, n_pts=2000, n_obs=9000, models=None, share=1):
    """A scene with every filter outcome present: gross feature errors, points pushed
    behind cameras, points pushed far away (tiny viewing angles), short tracks."""
    P = synth.make_problem(n_cam, n_pts, n_obs, seed=seed, scene="ring", spread=0.3, models=models,
                           shared_group_size=share, perturb=0.2)
    rng = np.random.default_rng(seed + 100)
    n = P.num_points
    bad_feat = rng.random(P.num_observations) < 0.03
    P.obs_xy[bad_feat] += rng.normal(0, 40.0, (int(bad_feat.sum()), 2))
    # far away points with CONSISTENT observations (zero reprojection error): only the
    # viewing-angle test can reject them
    far = rng.random(n) < 0.05
    P.points[far, :3] *= 400.0
    for o in np.flatnonzero(far[P.obs_point]):
        cam = P.obs_camera[o]
        g = P.camera_group[cam]
        K = P.intrinsics[P.group_offset[g]:P.group_offset[g + 1]]
        px, _ = oracle.project_point(int(P.group_model[g]), P.extrinsics[cam], K, P.points[P.obs_point[o]])
        if np.all(np.isfinite(px)):
            P.obs_xy[o] = px
    behind = rng.random(n) < 0.03
    P.points[behind, :3] *= -3.0
    return P


def tame_tracks(P):
    """Tracks all of whose observations are ordinary pixel coordinates.  A far-away point
    seen almost edge-on projects to ~1e11 px through the radial distortion polynomial; its
    squared error then carries rounding noise of order (1e-16 * 1e11)^2 px^2 in either
    implementation, so the VALUE of the mean is compared on the other tracks (flags always)."""
    big = np.abs(P.obs_xy).max(axis=1) > 1e5
    return np.bincount(P.obs_point[big], minlength=P.num_points) == 0


def check_flags(dev, ref, tame=None):
    flag_d, mean_d, fs = dev
    flag_o, mean_o, counts = ref
    np.testing.assert_array_equal(flag_d, flag_o)
    assert fs.num_estimated_tracks == counts[0]
    assert fs.num_bad_reprojections == counts[1]
    assert fs.num_insufficient_viewing_angles == counts[2]
    # a track that stops at a projection behind a camera holds a partial sum whose value
    # depends on the visiting order: compared on the other tracks only
    live = flag_o != 1
    ok = np.isfinite(mean_o) & live
    if tame is not None:
        ok &= tame
    np.testing.assert_allclose(mean_d[ok], mean_o[ok], rtol=1e-9, atol=1e-12)
    assert np.array_equal(np.isnan(mean_d[live]), np.isnan(mean_o[live]))


@pytest.mark.parametrize("models,share", [(None, 1), (MODELS, 1), (MODELS, 3)])
def test_filter_one_shot_matches_oracle(models, share):
    P = corrupted(21, models=models, share=share)
    ref = oracle.filter_outlier_tracks(P, 4.0, 2.0)
    assert len(np.unique(ref[0])) == 3, "the scene must contain all three outcomes"
    check_flags(lib.filter_outlier_tracks(P, 4.0, 2.0), ref, tame_tracks(P))
    # a second pair of thresholds moves tracks between the classes
    check_flags(lib.filter_outlier_tracks(P, 1.0, 8.0), oracle.filter_outlier_tracks(P, 1.0, 8.0),
                tame_tracks(P))


def test_filter_edge_cases():
    P = corrupted(22, n_cam=8, n_pts=150, n_obs=500)
    keep = np.ones(P.num_observations, bool)
    keep[P.obs_point == 0] = False                       # unobserved track
    keep[np.flatnonzero(P.obs_point == 1)[1:]] = False   # single-view track
    Q = P.copy()
    Q.obs_camera, Q.obs_point, Q.obs_xy = P.obs_camera[keep], P.obs_point[keep], P.obs_xy[keep]
    ref = oracle.filter_outlier_tracks(Q, 4.0, 2.0)
    assert ref[0][0] == 2 and ref[0][1] == 2
    check_flags(lib.filter_outlier_tracks(Q, 4.0, 2.0), ref, tame_tracks(Q))
    # no observations at all
    E = Q.copy()
    E.obs_camera, E.obs_point, E.obs_xy = Q.obs_camera[:0], Q.obs_point[:0], Q.obs_xy[:0]
    check_flags(lib.filter_outlier_tracks(E, 4.0, 2.0), oracle.filter_outlier_tracks(E, 4.0, 2.0))


def test_filter_on_resident_solver_after_ba_and_sharded():
    """BA -> filter with nothing re-uploaded: the filter sees the adjusted parameters; with
    the tracks sharded over two handles the per-rank flags tile the track set."""
    P = corrupted(23, n_cam=16, n_pts=800, n_obs=4000)
    o = abi.default_options(point_dof=3, max_num_iterations=5, loss_function_type=abi.LOSS_HUBER,
                            linear_solver_type=abi.ITERATIVE_SCHUR)
    s = lib.Solver(P.copy(), o)
    st, sm = s.solve(o)
    assert st == 0 and sm.success
    dev = s.filter_outlier_tracks(4.0, 2.0)
    adjusted = s.download()
    s.close()
    ref = oracle.filter_outlier_tracks(adjusted, 4.0, 2.0)
    check_flags(dev, ref, tame_tracks(adjusted))
    # sharded handles on the same (un-adjusted) parameters
    ref0 = oracle.filter_outlier_tracks(P, 4.0, 2.0)
    flag = np.full(P.num_points, 255, np.uint8)
    totals = np.zeros(3, np.int64)
    for r in range(2):
        sr = lib.Solver(P.copy(), o, rank=r, world=2)
        f, m, fs = sr.filter_outlier_tracks(4.0, 2.0)
        sr.close()
        own = f != 255
        assert not np.any(own & (flag != 255)), "a track was filtered by two ranks"
        flag[own] = f[own]
        totals += [fs.num_estimated_tracks, fs.num_bad_reprojections, fs.num_insufficient_viewing_angles]
    np.testing.assert_array_equal(flag, ref0[0])
    np.testing.assert_array_equal(totals, ref0[2])


@pytest.mark.parametrize("dof", [3, 4])
@pytest.mark.parametrize("loss", [abi.LOSS_TRIVIAL, abi.LOSS_HUBER, abi.LOSS_CAUCHY, abi.LOSS_SOFTLONE,
                                  abi.LOSS_ARCTAN, abi.LOSS_TUKEY])
def test_adjust_tracks_matches_oracle(dof, loss):
    P = synth.make_problem(20, 1500, 7000, seed=31 + dof, scene="ring", spread=0.3, models=MODELS)
    rng = np.random.default_rng(7)
    P.points[:, :3] += 0.5 * rng.standard_normal((P.num_points, 3))
    P.point_constant[::17] = 1
    o = abi.default_options(point_dof=dof, max_num_iterations=30, loss_function_type=loss,
                            robust_loss_width=3.0)
    R = P.copy()
    term_o, it_o, c0_o, c1_o = oracle.adjust_tracks(R, o)
    D = P.copy()
    term_d, it_d, c0_d, c1_d, ts = lib.adjust_tracks(D, o)
    np.testing.assert_array_equal(term_d, term_o)
    np.testing.assert_array_equal(it_d, it_o)
    np.testing.assert_allclose(c0_d, c0_o, rtol=1e-9, atol=1e-12)
    # point_dof = 4 (reference-exact homogeneous points) leaves the scale of X free: J X = 0,
    # the step along X is g_X / (clamped diagonal / radius) = rounding noise * 1e10, so two
    # correct implementations agree on the cost only to ~1e-5 relative and on X only up to
    # scale.  With point_dof = 3 the problems are well posed and the match is at round-off.
    ctol = 1e-9 if dof == 3 else 1e-4
    np.testing.assert_allclose(c1_d, c1_o, rtol=ctol, atol=1e-9)
    eucl = lambda X: X[:, :3] / X[:, 3:4]  # noqa: E731
    scale = np.abs(eucl(R.points)).max()
    assert np.abs(eucl(D.points) - eucl(R.points)).max() <= (1e-8 if dof == 3 else 1e-4) * scale
    adjusted = term_o >= 0
    assert ts.num_tracks == adjusted.sum() and ts.num_success == np.isin(term_o, (0, 1)).sum()
    assert ts.total_iterations == it_o[adjusted].sum()
    # constant / unobserved tracks are untouched, cameras never move
    np.testing.assert_array_equal(D.points[~adjusted], P.points[~adjusted])
    np.testing.assert_array_equal(D.extrinsics, P.extrinsics)
    np.testing.assert_array_equal(D.intrinsics, P.intrinsics)
    assert np.all(c1_d[adjusted] <= c0_d[adjusted] * (1 + 1e-12))


def test_adjust_tracks_iteration_limit_and_failed_start():
    P = synth.make_problem(10, 300, 1200, seed=41)
    rng = np.random.default_rng(3)
    P.points[:, :3] += 2.0 * rng.standard_normal((P.num_points, 3))
    # a track sitting exactly on a camera centre cannot be evaluated (reprojection_error.h:75-77)
    t = 5
    cam = P.obs_camera[np.flatnonzero(P.obs_point == t)[0]]
    P.points[t, :3] = P.extrinsics[cam, :3] * P.points[t, 3]
    o = abi.default_options(point_dof=3, max_num_iterations=2, function_tolerance=0.0,
                            parameter_tolerance=0.0, gradient_tolerance=0.0)
    R, D = P.copy(), P.copy()
    term_o, it_o, _, c1_o = oracle.adjust_tracks(R, o)
    term_d, it_d, _, c1_d, _ = lib.adjust_tracks(D, o)
    assert term_o[t] == 3 and (term_o == 1).sum() > 0
    np.testing.assert_array_equal(term_d, term_o)
    np.testing.assert_array_equal(it_d, it_o)
    np.testing.assert_array_equal(D.points[t], P.points[t])
    ok = term_o != 3
    np.testing.assert_allclose(c1_d[ok], c1_o[ok], rtol=1e-9, atol=1e-9)


def test_adjust_tracks_on_resident_solver_then_ba():
    """triangulate-ish -> per-track BA -> full BA on one resident handle."""
    P = synth.make_problem(12, 600, 2800, seed=43)
    rng = np.random.default_rng(5)
    P.points[:, :3] += 1.0 * rng.standard_normal((P.num_points, 3))
    o = abi.default_options(point_dof=3, max_num_iterations=20, linear_solver_type=abi.DENSE_SCHUR)
    s = lib.Solver(P.copy(), o)
    term, iters, c0, c1, ts = s.adjust_tracks(o)
    after_tracks = s.download().copy()
    R = P.copy()
    oracle.adjust_tracks(R, o)
    assert np.abs(after_tracks.points - R.points).max() <= 1e-8 * np.abs(R.points).max()
    st, sm = s.solve(o)
    assert st == 0 and sm.success
    s.close()
    st_o, sm_o = oracle.solve(R, o)
    assert sm.num_iterations == sm_o.num_iterations
    np.testing.assert_allclose(sm.final_cost, sm_o.final_cost, rtol=1e-9)


@pytest.mark.parametrize("models,share", [(None, 1), (MODELS, 3)])
@pytest.mark.parametrize("long_thr,cell,min_per_view", [(10, 100, 100), (3, 40, 20), (10, 100000, 50)])
def test_select_good_tracks_matches_oracle(models, share, long_thr, cell, min_per_view):
    """SelectGoodTracksForBundleAdjustment: device statistics + engine-side selection against
    the oracle's independent restatement; the selected sets must be identical."""
    P = synth.make_problem(24, 4000, 18000, seed=51, scene="ring", spread=0.3, models=models,
                           shared_group_size=share, perturb=0.5)
    sel_o, ln_o, err_o = oracle.select_good_tracks(P, long_thr, cell, min_per_view)
    sel_d, ln_d, err_d, ss = lib.select_good_tracks(P, long_thr, cell, min_per_view)
    np.testing.assert_array_equal(ln_d, ln_o)
    np.testing.assert_allclose(err_d, err_o, rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(sel_d, sel_o)
    assert ss.num_selected == sel_o.sum() and ss.num_tracks == P.num_points
    assert 0 < sel_o.sum() < P.num_points
    # restricted to a subset of the views
    mask = (np.arange(P.num_cameras) % 3 != 0).astype(np.uint8)
    sel_om, _, _ = oracle.select_good_tracks(P, long_thr, cell, min_per_view, view_mask=mask)
    sel_dm, _, _, _ = lib.select_good_tracks(P, long_thr, cell, min_per_view, view_mask=mask)
    np.testing.assert_array_equal(sel_dm, sel_om)


def test_select_then_partial_ba_on_resident_solver():
    """select -> statistics come from the resident parameters; sharded handles refuse."""
    P = synth.make_problem(12, 1200, 5000, seed=53)
    o = abi.default_options(point_dof=3, max_num_iterations=3)
    s = lib.Solver(P.copy(), o)
    sel_d, ln_d, err_d, ss = s.select_good_tracks(10, 100, 80)
    s.close()
    sel_o, ln_o, err_o = oracle.select_good_tracks(P, 10, 100, 80)
    np.testing.assert_array_equal(sel_d, sel_o)
    s2 = lib.Solver(P.copy(), o, rank=0, world=2)
    with pytest.raises(lib.EngineError):
        s2.select_good_tracks(10, 100, 80)
    s2.close()
