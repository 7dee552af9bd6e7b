"""-m gpu: compact planes (device_view.h, DeviceView::compact) against the stored camera block.

On the all-PINHOLE / default-mask / TRIVIAL-loss problem (every BAL problem, an unchanged Theia caller:
bundle_adjustment.h:95, pinhole_camera_model.h:86-94) with unit aspect ratio and zero skew, the 2 x 9 camera block of
ReprojectionError (reprojection_error.h:51-95) is a function of the point block, the normalised image point, the track and
the view; matrix-free iterations then store p_n (16 bytes) instead of the block (96 bytes) and the product / back-substitution
gather a transformed view vector.  TMI_BA_COMPACT_PLANES=0 keeps the stored block.

The two are the same operator evaluated in another order: identical LM / PCG iteration counts, costs and parameters to
round-off -- over both point parameterisations, long tracks (16 and 64 lanes per track, wavefront-per-track units), a view
with zero rotation (the first-order branch of AngleAxisRotatePoint), both SCHUR_JACOBI shapes, inner iterations, the
adaptive operator (an iteration that forms S re-linearizes with the full planes) and a sharded solve; and problems the
compact form does not cover (aspect ratio != 1, skew, another mask, another camera model) must not take it."""
import os

import numpy as np
import pytest

from theiasfm_amd import abi, lib, synth

pytestmark = pytest.mark.gpu

IMPL = dict(linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=abi.SCHUR_IMPLICIT)


def run(prob, compact, **kw):
    saved = {k: os.environ.pop(k, None) for k in ("TMI_BA_COMPACT_PLANES", "TMI_BA_MF_ONE_SWEEP")}
    try:
        os.environ["TMI_BA_COMPACT_PLANES"] = "1" if compact else "0"
        os.environ["TMI_BA_MF_ONE_SWEEP"] = "1"  # (the one-sweep product below its size threshold)
        p = prob.copy()
        kw.setdefault("max_num_iterations", 8)
        o = abi.default_options(use_inner_iterations=kw.pop("use_inner_iterations", 0), profile_kernels=1, **kw)
        trace = abi.attach_trace(o, kw["max_num_iterations"])
        st, s = lib.solve(p, o)
        assert st == 0, s.message
        return s, p, trace
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def same_trajectory(a, b, rtol=1e-7, trace_rtol=1e-8, atol=1e-9):
    sa, pa, ta = a
    sb, pb, tb = b
    assert sa.num_iterations == sb.num_iterations
    assert sa.num_successful_steps == sb.num_successful_steps and sa.num_unsuccessful_steps == sb.num_unsuccessful_steps
    assert sa.num_linear_solver_iterations == sb.num_linear_solver_iterations
    assert sa.termination == sb.termination
    assert abs(sa.final_cost - sb.final_cost) <= 1e-10 * abs(sa.final_cost)
    n = sa.num_iterations
    assert np.array_equal(ta[:n, 3], tb[:n, 3]) and np.array_equal(ta[:n, 6], tb[:n, 6])
    ok = np.isfinite(ta[:n]) & np.isfinite(tb[:n])
    np.testing.assert_allclose(ta[:n][ok], tb[:n][ok], rtol=trace_rtol, atol=0)
    # (parameters: PCG stops at eta = 0.1, so a round-off difference in the operator moves an iterate by ~1e-9 of its
    #  size -- more along the scale gauge of homogeneous points; the costs above agree to 1e-10)
    np.testing.assert_allclose(pa.points, pb.points, rtol=rtol, atol=atol)
    np.testing.assert_allclose(pa.extrinsics, pb.extrinsics, rtol=rtol, atol=atol)
    np.testing.assert_allclose(pa.intrinsics, pb.intrinsics, rtol=rtol, atol=atol)


def lin_us(s):
    i = abi.KERNEL_CLASS_NAMES.index("linearize")
    return s.kernel_seconds[i] / max(s.kernel_launches[i], 1)


def zero_rotation_view(prob):
    """view 0 with the identity rotation: theta^2 <= DBL_EPSILON takes the first-order branch (R = I, Jl = I)"""
    prob.extrinsics[0, 3:6] = 0.0
    sel = np.flatnonzero(prob.obs_camera == 0)
    prob.obs_xy[sel] = synth.project(prob, sel) + 0.3
    return prob


def tiny_rotation_view(prob):
    """view 0 with 0 < |w| <= 1.5e-8: the first-order branch with R = I + [w]x NOT orthogonal -- the one place where the
    compact rotation columns differ from the stored ones (by O(|w|) relative)"""
    prob.extrinsics[0, 3:6] = [1e-9, -2e-9, 1.5e-9]
    sel = np.flatnonzero(prob.obs_camera == 0)
    prob.obs_xy[sel] = synth.project(prob, sel) + 0.3
    return prob


CASES = {
    "tiny_rotation_view": (lambda: tiny_rotation_view(synth.make_problem(40, 6000, 32000, seed=47, scene="ring", spread=0.5)),
                           dict(point_dof=3, **IMPL)),
    "dof3": (lambda: synth.make_problem(60, 9000, 50000, seed=41, scene="ring", spread=0.4), dict(point_dof=3, **IMPL)),
    "dof4_parameter_blocks": (lambda: synth.make_problem(60, 9000, 50000, seed=43, scene="ring", spread=0.4),
                              dict(point_dof=4, preconditioner_type=abi.PRECOND_SCHUR_JACOBI_PARAMETER_BLOCKS, **IMPL)),
    "heavy_tail": (lambda: synth.make_problem(320, 30000, 190000, seed=7, scene="ring", spread=0.6, heavy_tail=0.01),
                   dict(point_dof=3, max_num_iterations=6, **IMPL)),
    "heavy_tail_dof4": (lambda: synth.make_problem(320, 30000, 190000, seed=8, scene="ring", spread=0.6, heavy_tail=0.01),
                        dict(point_dof=4, max_num_iterations=5, **IMPL)),
    "zero_rotation_view": (lambda: zero_rotation_view(synth.make_problem(40, 6000, 32000, seed=45, scene="ring", spread=0.5)),
                           dict(point_dof=3, **IMPL)),
    "identity_preconditioner": (lambda: synth.make_problem(40, 6000, 32000, seed=46, scene="ring", spread=0.5),
                                dict(point_dof=3, preconditioner_type=abi.PRECOND_IDENTITY, **IMPL)),
    "inner_iterations": (lambda: synth.make_problem(40, 5000, 28000, seed=57, scene="ring", spread=0.5),
                         dict(point_dof=4, use_inner_iterations=1, max_num_iterations=5, **IMPL)),
    "rejected_steps": (lambda: synth.make_problem(40, 5000, 28000, seed=55, scene="ring", spread=0.5),
                       dict(point_dof=3, initial_trust_region_radius=1e12, max_num_iterations=10, **IMPL)),
    "auto": (lambda: synth.make_problem(50, 7000, 40000, seed=49, scene="ring", spread=0.5),
             dict(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=abi.SCHUR_AUTO)),
    # robust losses (the reference's application flags ask for HUBER, applications/build_reconstruction_flags.txt:117): the
    # planes then hold the corrected point C p_n and r^2 of the uncorrected one (DeviceView::compact == 2)
    "huber": (lambda: synth.make_problem(40, 6000, 32000, seed=61, scene="ring", spread=0.5, heavy_tail=0.01),
              dict(point_dof=3, loss_function_type=abi.LOSS_HUBER, robust_loss_width=2.0, **IMPL)),
    "cauchy_dof4_heavy_tail": (lambda: synth.make_problem(320, 30000, 190000, seed=9, scene="ring", spread=0.6, heavy_tail=0.01),
                               dict(point_dof=4, loss_function_type=abi.LOSS_CAUCHY, robust_loss_width=3.0, max_num_iterations=5, **IMPL)),
    "tukey": (lambda: synth.make_problem(40, 6000, 32000, seed=66, scene="ring", spread=0.5),
              dict(point_dof=3, loss_function_type=abi.LOSS_TUKEY, robust_loss_width=20.0, **IMPL)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_compact_planes_walk_the_same_trajectory(name):
    make, kw = CASES[name]
    prob = make()
    a = run(prob, False, **dict(kw))
    b = run(prob, True, **dict(kw))
    assert a[0].num_linear_solver_iterations > 0
    # (IDENTITY: PCG on the unpreconditioned reduced system amplifies round-off in the operator by its condition number;
    #  the model cost change and step norm of the late iterations then agree to ~1e-5, the costs still to 1e-10)
    loose = kw.get("preconditioner_type") == abi.PRECOND_IDENTITY
    same_trajectory(a, b, rtol=1e-4 if loose else 1e-7, trace_rtol=1e-3 if loose else 1e-8, atol=1e-5 if loose else 1e-9)


def test_landmark_tracks_take_the_wavefront_per_track_units():
    """tracks that every one of 560 views sees: the product's wide path (u and t through the global scratch)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("one_sweep_cases", os.path.join(os.path.dirname(__file__), "test_gpu_one_sweep.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with_landmarks = mod.with_landmarks
    prob = with_landmarks(synth.make_problem(560, 12000, 70000, seed=13, scene="ring", spread=0.5, heavy_tail=0.01), 6, 14)
    kw = dict(point_dof=3, max_num_iterations=5, **IMPL)
    same_trajectory(run(prob, False, **dict(kw)), run(prob, True, **dict(kw)))


def test_compact_planes_are_taken_and_cheaper_to_write():
    """the specialised linearize stores 80 instead of 160 bytes per observation"""
    prob = synth.make_problem(200, 60000, 400000, seed=3, scene="ring", spread=0.4)
    kw = dict(point_dof=3, max_num_iterations=4, **IMPL)
    a = run(prob, False, **dict(kw))[0]
    b = run(prob, True, **dict(kw))[0]
    assert lin_us(b) < 0.95 * lin_us(a)


@pytest.mark.parametrize("name", ["aspect_ratio", "mask", "radtan"])
def test_problems_outside_the_compact_form_keep_the_full_planes(name):
    """the switch must change nothing -- bit for bit -- where the compact form does not apply"""
    kw = dict(point_dof=3, **IMPL)
    if name == "aspect_ratio":
        prob = synth.make_problem(40, 6000, 32000, seed=62, scene="ring", spread=0.5)
        prob.intrinsics[1] = 1.02  # the first view's aspect ratio (constant under the default mask)
        sel = np.flatnonzero(prob.obs_camera == 0)
        prob.obs_xy[sel] = synth.project(prob, sel) + 0.3
    elif name == "mask":
        prob = synth.make_problem(40, 6000, 32000, seed=63, scene="ring", spread=0.5,
                                  intrinsics_to_optimize=abi.INTRINSICS_FOCAL_LENGTH | abi.INTRINSICS_PRINCIPAL_POINTS)
    else:
        prob = synth.make_problem(30, 4000, 22000, seed=64, scene="ring", spread=0.5, models=[(abi.PINHOLE_RADIAL_TANGENTIAL, 1.0)])
    a = run(prob, False, **dict(kw))
    b = run(prob, True, **dict(kw))
    assert a[0].final_cost == b[0].final_cost and np.array_equal(a[2], b[2], equal_nan=True)
    assert np.array_equal(a[1].points, b[1].points) and np.array_equal(a[1].extrinsics, b[1].extrinsics)


def test_skewed_view_keeps_the_full_planes_and_matches_the_oracle_minimum():
    """skew != 0 on one view: the handle must fall back (a compact evaluation would drop the skew term of the k1, k2 columns)"""
    prob = synth.make_problem(30, 4000, 22000, seed=65, scene="ring", spread=0.5)
    prob.intrinsics[2] = 0.01 * prob.intrinsics[0]
    sel = np.flatnonzero(prob.obs_camera == 0)
    prob.obs_xy[sel] = synth.project(prob, sel) + 0.3
    kw = dict(point_dof=3, **IMPL)
    a = run(prob, False, **dict(kw))
    b = run(prob, True, **dict(kw))
    assert a[0].final_cost == b[0].final_cost and np.array_equal(a[1].points, b[1].points)


@pytest.mark.parametrize("compact", [False, True])
def test_operator_info_reports_the_plane_layout(compact):
    """tmi_ba_solver_operator_info out[7]: the planes of the handle's last linearisation (bench.py's layout floor reads it)"""
    prob = synth.make_problem(40, 6000, 32000, seed=71, scene="ring", spread=0.5)
    saved = {k: os.environ.pop(k, None) for k in ("TMI_BA_COMPACT_PLANES", "TMI_BA_MF_ONE_SWEEP")}
    try:
        os.environ["TMI_BA_COMPACT_PLANES"] = "1" if compact else "0"
        os.environ["TMI_BA_MF_ONE_SWEEP"] = "1"
        o = abi.default_options(max_num_iterations=3, use_inner_iterations=0, point_dof=3, **IMPL)
        s = lib.Solver(prob.copy(), o, 0, 1)
        try:
            st, sm = s.solve(o)
            assert st == 0, sm.message
            assert s.operator_info()["compact_planes"] == compact
            # the Jacobian the caller can ask for is always the full one, and asking resets the flag
            s.evaluate(3)
            assert s.operator_info()["compact_planes"] is False
        finally:
            s.close()
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def test_an_iteration_that_forms_S_relinearizes_with_the_full_planes():
    """schur_mode auto with a low break-even (TMI_BA_BREAK_EVEN): the first LM iterations are matrix-free on compact planes,
    a later one forms S -- its records need the stored camera block, so the engine linearises again with the full planes
    (engine.hip, `v.compact && !v.direct_diag`); same trajectory as the run that never used compact planes"""
    prob = synth.make_problem(50, 7000, 40000, seed=49, scene="ring", spread=0.5)
    kw = dict(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=abi.SCHUR_AUTO, max_num_iterations=6,
              function_tolerance=-1.0, gradient_tolerance=-1.0, parameter_tolerance=-1.0)  # (six: before the iterates are round-off)
    saved = os.environ.pop("TMI_BA_BREAK_EVEN", None)
    try:
        os.environ["TMI_BA_BREAK_EVEN"] = "2"
        a = run(prob, False, **dict(kw))
        b = run(prob, True, **dict(kw))
    finally:
        os.environ.pop("TMI_BA_BREAK_EVEN", None)
        if saved is not None:
            os.environ["TMI_BA_BREAK_EVEN"] = saved
    mf = b[0].num_matrix_free_iterations
    assert 0 < mf < b[0].num_iterations, (mf, b[0].num_iterations)  # both operators ran
    same_trajectory(a, b)


def with_constant_views(prob, every):
    """every `every`-th view constant altogether (extrinsics and its private intrinsics): the views outside the subset of
    BundleAdjustPartialReconstruction (bundle_adjuster.cc:141-180) -- they have no block in the reduced system"""
    for c in range(0, prob.num_cameras, every):
        prob.camera_flags[c] = abi.CAMERA_POSITION_CONSTANT | abi.CAMERA_ORIENTATION_CONSTANT
        g = prob.camera_group[c]
        prob.intrinsics_constant[prob.group_offset[g]:prob.group_offset[g + 1]] = 1
    return prob


@pytest.mark.parametrize("dof", [3, 4])
def test_partial_adjustment_with_constant_views_takes_the_compact_planes(dof):
    """a third of the views constant: the specialised bodies and the compact planes serve the free ones (the constant
    views' observations still feed the point side); same trajectory as the stored block, same minimum as the oracle"""
    from oracle import oracle
    prob = with_constant_views(synth.make_problem(60, 9000, 50000, seed=81, scene="ring", spread=0.4), 3)
    kw = dict(point_dof=dof, max_num_iterations=6, **IMPL)
    a = run(prob, False, **dict(kw))
    b = run(prob, True, **dict(kw))
    same_trajectory(a, b)
    saved = os.environ.pop("TMI_BA_COMPACT_PLANES", None)
    o = abi.default_options(use_inner_iterations=0, **kw)
    ref = prob.copy()
    st_o, s_o = oracle.solve(ref, o)
    assert st_o == 0
    assert abs(b[0].final_cost - s_o.final_cost) <= 1e-9 * s_o.final_cost
    assert b[0].num_iterations == s_o.num_iterations and b[0].num_linear_solver_iterations == s_o.num_linear_solver_iterations
    # the constant views did not move
    fixed = np.arange(0, prob.num_cameras, 3)
    assert np.array_equal(b[1].extrinsics[fixed], prob.extrinsics[fixed])
    if saved is not None:
        os.environ["TMI_BA_COMPACT_PLANES"] = saved
