"""CPU: the oracle's restatement of the post-BA outlier filter and of the per-track BA
(SURVEY 8(f) rows 1 and 3), pinned by the reference's own tests where it has them:

  * SufficientTriangulationAngle: triangulation_test.cc:432-497 restated case by case;
  * SetOutlierTracksToUnestimated has no reference test -- its three outcomes are
    exercised on hand-built tracks whose expected flags follow from the reference code
    (set_outlier_tracks_to_unestimated.cc:62-133) by inspection.
"""
import numpy as np

from oracle import oracle
from theiasfm_amd import abi, synth


def circle_rays(n, step_deg):
    a = np.deg2rad(np.arange(n) * step_deg)
    return np.stack([np.cos(a), np.sin(a), np.zeros(n)], 1)


def test_sufficient_angle_all_sufficient():
    # triangulation_test.cc:432-449
    for i in range(2, 50):
        assert oracle.sufficient_triangulation_angle(circle_rays(i, 5.0), 4.0)


def test_sufficient_angle_all_insufficient():
    # triangulation_test.cc:451-468
    for i in range(2, 50):
        assert not oracle.sufficient_triangulation_angle(circle_rays(i, 4.0 / (i + 1e-4)), 4.0)


def test_sufficient_angle_some_and_two():
    # triangulation_test.cc:470-497
    a = np.deg2rad([0.0, 5.0, 1.0])
    rays = np.stack([np.cos(a), np.sin(a), np.zeros(3)], 1)
    assert oracle.sufficient_triangulation_angle(rays, 4.0)
    assert not oracle.sufficient_triangulation_angle(rays[[0, 2]], 4.0)
    assert not oracle.sufficient_triangulation_angle(rays[:1], 4.0)
    assert not oracle.sufficient_triangulation_angle(np.zeros((0, 3)), 4.0)


def two_camera_problem(points, baseline=10.0, noise=None):
    """Two identity-rotation pinhole cameras at (0,0,0) and (baseline,0,0) looking down +z."""
    pts = np.asarray(points, dtype=np.float64)
    n = len(pts)
    K = np.array([800.0, 1.0, 0.0, 500.0, 400.0, 0.0, 0.0])
    ext = np.zeros((2, 6))
    ext[1, 0] = baseline
    cam = np.tile(np.arange(2, dtype=np.int32), n)
    pt = np.repeat(np.arange(n, dtype=np.int32), 2)
    X = np.concatenate([pts, np.ones((n, 1))], 1)
    xy = np.zeros((2 * n, 2))
    for o in range(2 * n):
        px, _ = oracle.project_point(abi.PINHOLE, ext[cam[o]], K, X[pt[o]])
        xy[o] = px
    if noise is not None:
        xy += noise
    return abi.Problem(extrinsics=ext, camera_group=np.zeros(2, np.int32),
                       camera_flags=np.zeros(2, np.uint8), group_model=np.array([abi.PINHOLE], np.int32),
                       group_offset=np.array([0, 7], np.int32), intrinsics=K.copy(),
                       intrinsics_constant=np.zeros(7, np.uint8), points=X,
                       point_constant=np.zeros(n, np.uint8), obs_camera=cam, obs_point=pt, obs_xy=xy)


def test_filter_outcomes_by_construction():
    # track 0: well conditioned, exact observations -> kept
    # track 1: far away (viewing angle ~0.06 deg) -> insufficient angle
    # track 2: well conditioned, 5 px error on both observations -> mean sq error 25 > 16
    # track 3: behind both cameras -> bad reprojection (depth < 0)
    pts = [[5.0, 0.0, 50.0], [5.0, 0.0, 10000.0], [4.0, 1.0, 40.0], [5.0, 0.0, -30.0]]
    noise = np.zeros((8, 2))
    noise[4:6, 0] = 5.0
    P = two_camera_problem(pts, noise=noise)
    flag, mean, counts = oracle.filter_outlier_tracks(P, 4.0, 2.0)
    assert flag.tolist() == [0, 2, 1, 1]
    assert counts.tolist() == [4, 2, 1]
    np.testing.assert_allclose(mean[:3], [0.0, 0.0, 25.0], atol=1e-18, rtol=1e-12)
    # the threshold is strict (mean > max^2): exactly at the threshold the track stays
    flag2, _, _ = oracle.filter_outlier_tracks(P, 5.0, 2.0)
    assert flag2[2] == 0
    # negative homogeneous scale flips the sign of the depth (rotated_z / w, camera.cc:212)
    P.points[0] *= -1.0
    flag3, _, _ = oracle.filter_outlier_tracks(P, 4.0, 2.0)
    assert flag3[0] == 0  # same point, same rays (hnormalized), depth z/w unchanged in sign: (-z)/(-1)


def test_filter_unobserved_and_single_view_tracks():
    P = synth.make_problem(6, 40, 160, seed=5)
    # drop all observations of track 0 and all but one of track 1
    keep = np.ones(P.num_observations, bool)
    keep[P.obs_point == 0] = False
    idx1 = np.flatnonzero(P.obs_point == 1)
    keep[idx1[1:]] = False
    Q = P.copy()
    Q.obs_camera, Q.obs_point, Q.obs_xy = P.obs_camera[keep], P.obs_point[keep], P.obs_xy[keep]
    flag, mean, counts = oracle.filter_outlier_tracks(Q, 1e6, 0.0)
    # no pair of rays -> SufficientTriangulationAngle is false (triangulation.cc:242-249)
    assert flag[0] == 2 and flag[1] == 2
    assert np.isnan(mean[0])  # 0 / 0 as in the reference (:108)
    assert counts[0] == Q.num_points


def test_adjust_tracks_matches_single_track_solve():
    """oracle_adjust_tracks is oracle_ba_solve on one-track sub-problems: check one track
    against an explicit sub-problem built here, and that the cost never increases."""
    P = synth.make_problem(8, 60, 300, seed=11)
    rng = np.random.default_rng(1)
    P.points[:, :3] += 0.3 * rng.standard_normal((P.num_points, 3))
    o = abi.default_options(point_dof=4, max_num_iterations=25)
    Q = P.copy()
    term, iters, c0, c1 = oracle.adjust_tracks(Q, o)
    assert set(np.unique(term)) <= {0, 1}
    assert np.all(c1 <= c0 * (1 + 1e-12))
    assert np.all(iters >= 1)
    t = 7
    sel = P.obs_point == t
    S = P.copy()
    S.camera_flags = np.full(P.num_cameras, abi.CAMERA_POSITION_CONSTANT | abi.CAMERA_ORIENTATION_CONSTANT, np.uint8)
    S.intrinsics_constant = np.ones_like(P.intrinsics_constant)
    S.points = P.points[t:t + 1].copy()
    S.point_constant = np.zeros(1, np.uint8)
    S.obs_camera, S.obs_xy = P.obs_camera[sel], P.obs_xy[sel]
    S.obs_point = np.zeros(int(sel.sum()), np.int32)
    # BundleAdjustTrack: DENSE_QR, no inner iterations (bundle_adjustment.cc:100-101)
    o2 = abi.default_options(point_dof=4, max_num_iterations=25, linear_solver_type=abi.DENSE_QR,
                             use_inner_iterations=0)
    st, sm = oracle.solve(S, o2)
    assert st == 0
    np.testing.assert_allclose(Q.points[t], S.points[0], rtol=1e-12, atol=1e-12)
    assert sm.num_iterations == iters[t]
    np.testing.assert_allclose(sm.final_cost, c1[t], rtol=1e-12)


# ---- SelectGoodTracksForBundleAdjustment (select_good_tracks_for_bundle_adjustment.cc) ----
def test_select_good_tracks_rules_by_construction():
    """Two views, tracks laid out so that every rule of the reference code is visible."""
    # tracks 0..5 seen by both cameras; feature positions chosen per cell (cell size 100 px)
    pts = [[1.0, 0.0, 50.0], [1.2, 0.0, 50.0], [6.0, 3.0, 50.0], [6.2, 3.0, 50.0], [-4.0, -3.0, 60.0],
           [-4.1, -3.0, 60.0]]
    P = two_camera_problem(pts, baseline=10.0)
    # give every track a distinct, known mean error: shift its features in x by d pixels
    d = np.array([3.0, 1.0, 2.0, 4.0, 0.5, 0.6])
    P.obs_xy[:, 0] += np.repeat(d, 2)
    sel, ln, err = oracle.select_good_tracks(P, 10, 100, 0)
    assert ln.tolist() == [2] * 6
    np.testing.assert_allclose(err, d * d, rtol=1e-9)
    cells = {}
    for cam in range(2):
        for o in np.flatnonzero(P.obs_camera == cam):
            key = (cam, int(P.obs_xy[o, 0] / 100.0), int(P.obs_xy[o, 1] / 100.0))
            cells.setdefault(key, []).append(int(P.obs_point[o]))
    expect = set()
    for members in cells.values():
        expect.add(min(members, key=lambda t: (ln[t], err[t], t)))
    assert set(np.flatnonzero(sel)) == expect
    assert 0 < len(expect) < 6
    # the (truncated) length is the FIRST key and the minimum wins (:65-69, :187-190): a track
    # seen by one view only beats every two-view track of its cell whatever its error
    keep = np.ones(P.num_observations, bool)
    t_short = max(expect ^ set(range(6)))  # a track that lost its cell
    keep[np.flatnonzero((P.obs_point == t_short) & (P.obs_camera == 1))] = False
    Q = P.copy()
    Q.obs_camera, Q.obs_point, Q.obs_xy = P.obs_camera[keep], P.obs_point[keep], P.obs_xy[keep]
    sel2, ln2, _ = oracle.select_good_tracks(Q, 10, 100, 0)
    assert ln2[t_short] == 1 and sel2[t_short] == 1
    # truncation: with threshold 1 every length ties and the error decides
    sel3, ln3, _ = oracle.select_good_tracks(Q, 1, 100, 0)
    assert ln3.max() == 1
    # top-up: with a huge cell nothing but one track per view is chosen by the grid step, the
    # rest is filled in ascending track index (:236-247) up to the minimum
    sel4, _, _ = oracle.select_good_tracks(P, 10, 100000, 4)
    grid_only, _, _ = oracle.select_good_tracks(P, 10, 100000, 0)
    assert grid_only.sum() <= 2
    chosen = set(np.flatnonzero(sel4))
    assert len(chosen) == 4
    rest = sorted(set(range(6)) - set(np.flatnonzero(grid_only)))
    assert chosen == set(np.flatnonzero(grid_only)) | set(rest[:4 - int(grid_only.sum())])
    # a view mask removes that view's cells and top-up but not its observations from the statistics
    sel5, ln5, err5 = oracle.select_good_tracks(P, 10, 100, 0, view_mask=[1, 0])
    assert ln5.tolist() == [2] * 6
    assert set(np.flatnonzero(sel5)) <= expect


def test_select_good_tracks_min_per_view_is_met():
    P = synth.make_problem(10, 1500, 7000, seed=17)
    sel, ln, err = oracle.select_good_tracks(P, 10, 100, 120)
    per_view = np.bincount(P.obs_camera[sel[P.obs_point] == 1], minlength=P.num_cameras)
    seen = np.bincount(P.obs_camera, minlength=P.num_cameras)
    assert np.all(per_view >= np.minimum(120, seen))
    assert sel.sum() < P.num_points
