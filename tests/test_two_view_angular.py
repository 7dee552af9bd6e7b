"""BundleAdjustTwoViewsAngular (bundle_adjust_two_views.cc:193-240) for a batch of view pairs.

The reference has no test of this function and nothing in its tree calls it; what can be pinned is
what its pieces are by construction: AngularEpipolarError (angular_epipolar_error.h:47-89) vanishes
at the true relative pose of exact correspondences, UnitNormThreeVectorParameterization keeps the
position on the unit sphere, the solve decreases the cost, and the device agrees with the oracle
(dual numbers on both sides, written independently in C and in HIP)."""
import numpy as np
import pytest

from oracle import oracle
from theiasfm_amd import synth


def test_angular_epipolar_error_vanishes_at_the_true_pose():
    B, rot, pos = synth.make_two_view_angular_batch(5, 11, noise=0.0)
    for p in range(B.num_pairs):
        for q in range(B.correspondence_ptr[p], B.correspondence_ptr[p] + 5):
            e = oracle.angular_epipolar_error(rot[p], pos[p], B.features1[q], B.features2[q])
            assert e is not None and abs(e) < 1e-14
            e0 = oracle.angular_epipolar_error(B.rotation2[p], B.position2[p], B.features1[q], B.features2[q])
            assert e0 is not None and e0 >= 0.0  # a / 2 - sqrt(a^2 / 4 - b^2) >= 0 while a >= 0


def test_oracle_solve_decreases_the_cost_and_keeps_the_position_on_the_sphere():
    B, rot, pos = synth.make_two_view_angular_batch(60, 4, noise=1e-3)
    C = B.copy()
    term, it, c0, c1 = oracle.adjust_two_views_angular(C)
    assert (term == 0).all() and (it > 0).all() and (it < 200).all(), (term, it)
    assert (c1 <= c0).all() and np.median(c1 / c0) < 0.2
    np.testing.assert_allclose(np.linalg.norm(C.position2, axis=1), 1.0, atol=1e-14)
    # exact data: the cost goes to (numerically) nothing
    B0, _, _ = synth.make_two_view_angular_batch(20, 5, noise=0.0)
    C0 = B0.copy()
    term, it, c0, c1 = oracle.adjust_two_views_angular(C0)
    assert (c1 < 1e-6 * c0).all()


def test_a_pair_without_correspondences_is_left_alone():
    B, _, _ = synth.make_two_view_angular_batch(3, 6)
    ptr = B.correspondence_ptr.copy()
    ptr[2] = ptr[1]  # pair 1 empty
    B.correspondence_ptr = ptr
    C = B.copy()
    term, it, c0, c1 = oracle.adjust_two_views_angular(C)
    assert term[1] == -1 and it[1] == 0
    np.testing.assert_array_equal(C.rotation2[1], B.rotation2[1])


@pytest.mark.gpu
def test_device_matches_oracle():
    from theiasfm_amd import lib
    B, _, _ = synth.make_two_view_angular_batch(400, 9, noise=1e-3)
    ptr = B.correspondence_ptr.copy()
    ptr[8] = ptr[7]  # an empty pair
    B.correspondence_ptr = ptr
    D, O = B.copy(), B.copy()
    term_d, it_d, c0_d, c1_d, ts = lib.adjust_two_views_angular(D)
    term_o, it_o, c0_o, c1_o = oracle.adjust_two_views_angular(O)
    np.testing.assert_array_equal(term_d, term_o)
    np.testing.assert_allclose(c0_d, c0_o, rtol=1e-9, atol=1e-30)
    # the residual is a squared angle: costs of 1e-12 and a flat minimum; iteration counts may differ by
    # one where a tolerance test is decided by the last bits
    assert (np.abs(it_d - it_o) <= 1).all() and (it_d == it_o).mean() > 0.97
    same = it_d == it_o
    np.testing.assert_allclose(c1_d[same], c1_o[same], rtol=1e-6, atol=1e-30)
    np.testing.assert_allclose(D.rotation2[same], O.rotation2[same], atol=1e-7)
    np.testing.assert_allclose(D.position2[same], O.position2[same], atol=1e-7)
    assert ts.num_tracks == B.num_pairs - 1 and ts.num_success == int(((term_o == 0) | (term_o == 1)).sum())
