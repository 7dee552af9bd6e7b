"""-m gpu: the camera side of a matrix-free LM iteration built WITHOUT camera-major records (direct_diag.h, round 5)
against the record path (TMI_BA_DIRECT_DIAG=0: point_eliminate writes [A | Q] + tail per observation, camera_diag
reads them back).

What is computed is the f-block diagonal of Ceres' SchurEliminator (under ceres::Solve, bundle_adjuster.cc:205):
S_cc = sum A^T (I - Q Q^T) A, the U diagonal, g~ and g_c of every view.  The direct path re-evaluates every observation
view by view (reprojection_error.h:51-95 on the view's prepared record, the loss corrector, the Jacobi scales) from a
128-byte record per TRACK; the sums must give the record path's trajectory to round-off whatever the camera model, the
loss, the point parameterisation, the block width, constant points / constant camera positions (position columns
stored) -- and the oracle's."""
import os

import numpy as np
import pytest

from theiasfm_amd import abi, lib, synth

pytestmark = pytest.mark.gpu

ENV = ("TMI_BA_DIRECT_DIAG", "TMI_BA_MF_ONE_SWEEP", "TMI_BA_SETUP_TIMING", "TMI_BA_COST_BY_VIEW")


def run(prob, direct, cost_by_view=True, **kw):
    saved = {k: os.environ.pop(k, None) for k in ENV}
    try:
        if not direct:
            os.environ["TMI_BA_DIRECT_DIAG"] = "0"
        if not cost_by_view:
            os.environ["TMI_BA_COST_BY_VIEW"] = "0"
        os.environ["TMI_BA_MF_ONE_SWEEP"] = "1"  # (the one-sweep product below its size threshold)
        p = prob.copy()
        kw.setdefault("max_num_iterations", 6)
        st, s = lib.solve(p, abi.default_options(use_inner_iterations=0, **kw))
        assert st == 0, s.message
        return s, p
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


IMPL = dict(linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=abi.SCHUR_IMPLICIT)


def _const_point(p):
    p = p.copy()
    p.point_constant[5] = 1
    p.point_constant[77] = 1
    return p


def _const_position(p):
    p = p.copy()
    p.camera_flags[3] = abi.CAMERA_POSITION_CONSTANT
    p.camera_flags[11] = abi.CAMERA_ORIENTATION_CONSTANT
    return p


CASES = {
    "pinhole_dof3": (lambda: synth.make_problem(60, 9000, 50000, seed=31, scene="ring", spread=0.4), dict(point_dof=3, **IMPL)),
    "cauchy_heavy_tail": (lambda: synth.make_problem(320, 20000, 120000, seed=23, scene="ring", spread=0.6, heavy_tail=0.01),
                          dict(point_dof=3, loss_function_type=abi.LOSS_CAUCHY, robust_loss_width=3.0, **IMPL)),
    "huber_dof4": (lambda: synth.make_problem(40, 6000, 30000, seed=21, scene="ring", spread=0.5),
                   dict(point_dof=4, loss_function_type=abi.LOSS_HUBER, robust_loss_width=2.0, **IMPL)),
    "d6_dof4": (lambda: synth.make_problem(60, 9000, 50000, seed=25, scene="ring", spread=0.4,
                                           intrinsics_to_optimize=abi.INTRINSICS_NONE), dict(point_dof=4, **IMPL)),
    "mixed_models": (lambda: synth.make_problem(48, 6000, 36000, seed=33, scene="ring", spread=0.5,
                                                models=[(abi.PINHOLE, 0.2), (abi.PINHOLE_RADIAL_TANGENTIAL, 0.2), (abi.FISHEYE, 0.2),
                                                        (abi.FOV, 0.2), (abi.DIVISION_UNDISTORTION, 0.2)],
                                                intrinsics_to_optimize=abi.INTRINSICS_FOCAL_LENGTH | abi.INTRINSICS_PRINCIPAL_POINTS),
                     dict(point_dof=3, loss_function_type=abi.LOSS_SOFTLONE, robust_loss_width=2.0, **IMPL)),
    "d12_radtan": (lambda: synth.make_problem(30, 4000, 22000, seed=27, scene="ring", spread=0.5,
                                              models=[(abi.PINHOLE_RADIAL_TANGENTIAL, 1.0)],
                                              intrinsics_to_optimize=abi.INTRINSICS_FOCAL_LENGTH | abi.INTRINSICS_PRINCIPAL_POINTS
                                              | abi.INTRINSICS_RADIAL_DISTORTION), dict(point_dof=3, **IMPL)),
    "constant_points": (lambda: _const_point(synth.make_problem(40, 5000, 28000, seed=35, scene="ring", spread=0.5)),
                        dict(point_dof=3, **IMPL)),
    "constant_position_and_rotation": (lambda: _const_position(synth.make_problem(40, 5000, 28000, seed=37, scene="ring", spread=0.5)),
                                       dict(point_dof=4, loss_function_type=abi.LOSS_HUBER, robust_loss_width=2.0, **IMPL)),
    "no_jacobi_scaling": (lambda: synth.make_problem(40, 5000, 28000, seed=39, scene="ring", spread=0.5),
                          dict(point_dof=3, jacobi_scaling=0, **IMPL)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_direct_camera_side_gives_the_record_paths_trajectory(name):
    make, kw = CASES[name]
    prob = make()
    s1, p1 = run(prob, True, **kw)
    s0, p0 = run(prob, False, **kw)
    assert s1.num_iterations == s0.num_iterations and s1.num_successful_steps == s0.num_successful_steps
    assert s1.num_linear_solver_iterations == s0.num_linear_solver_iterations
    assert abs(s1.initial_cost - s0.initial_cost) <= 1e-13 * s0.initial_cost
    assert abs(s1.final_cost - s0.final_cost) <= 1e-10 * s0.final_cost
    scale = max(1.0, np.abs(p0.extrinsics).max())
    assert np.abs(p1.extrinsics - p0.extrinsics).max() <= 1e-8 * scale
    assert np.abs(p1.intrinsics - p0.intrinsics).max() <= 1e-8 * max(1.0, np.abs(p0.intrinsics).max())
    assert np.abs(p1.points - p0.points).max() <= 1e-8 * max(1.0, np.abs(p0.points).max())


def test_direct_camera_side_follows_the_oracle():
    from oracle import oracle
    prob = synth.make_problem(36, 4000, 24000, seed=41, scene="ring", spread=0.5,
                              models=[(abi.PINHOLE, 0.5), (abi.FISHEYE, 0.5)])
    kw = dict(point_dof=3, loss_function_type=abi.LOSS_HUBER, robust_loss_width=2.0, max_num_iterations=5, **IMPL)
    s1, p1 = run(prob, True, **kw)
    ref = prob.copy()
    st, s2 = oracle.solve(ref, abi.default_options(use_inner_iterations=0, **kw))
    assert st == 0
    assert s1.num_iterations == s2.num_iterations and s1.num_linear_solver_iterations == s2.num_linear_solver_iterations
    assert abs(s1.final_cost - s2.final_cost) <= 1e-9 * s2.final_cost


def test_auto_mode_switches_between_the_paths_per_iteration():
    # schur_mode auto: short PCG solves run matrix-free (direct camera side), long ones on the formed S (records)
    prob = synth.make_problem(120, 16000, 90000, seed=43, scene="ring", spread=0.5)
    kw = dict(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=abi.SCHUR_AUTO, max_num_iterations=8)
    saved = os.environ.pop("TMI_BA_BREAK_EVEN", None)
    try:
        os.environ["TMI_BA_BREAK_EVEN"] = "3"
        s1, p1 = run(prob, True, **kw)
        s0, p0 = run(prob, False, **kw)
    finally:
        os.environ.pop("TMI_BA_BREAK_EVEN", None)
        if saved is not None:
            os.environ["TMI_BA_BREAK_EVEN"] = saved
    assert s1.num_iterations == s0.num_iterations
    assert s1.num_linear_solver_iterations == s0.num_linear_solver_iterations
    assert abs(s1.final_cost - s0.final_cost) <= 1e-10 * s0.final_cost


@pytest.mark.parametrize("name", ["cauchy_heavy_tail", "mixed_models", "huber_dof4"])
def test_trial_cost_view_by_view_gives_the_track_major_one(name):
    # ddg::cost_view_kernel against cost_kernel (TMI_BA_COST_BY_VIEW=0): the same residuals summed in another order
    make, kw = CASES[name]
    prob = make()
    s1, p1 = run(prob, True, True, **kw)
    s0, p0 = run(prob, True, False, **kw)
    assert s1.num_iterations == s0.num_iterations and s1.num_successful_steps == s0.num_successful_steps
    assert s1.num_linear_solver_iterations == s0.num_linear_solver_iterations
    assert abs(s1.final_cost - s0.final_cost) <= 1e-12 * s0.final_cost
    assert abs(s1.final_rmse - s0.final_rmse) <= 1e-12 * s0.final_rmse


def test_a_fully_constant_camera_keeps_the_track_major_cost():
    # its observations own no camera-major slot: the view-by-view cost would not see them
    prob = synth.make_problem(40, 5000, 28000, seed=45, scene="ring", spread=0.5,
                              intrinsics_to_optimize=abi.INTRINSICS_NONE)
    prob.camera_flags[7] = abi.CAMERA_POSITION_CONSTANT | abi.CAMERA_ORIENTATION_CONSTANT
    kw = dict(point_dof=3, **IMPL)
    s1, p1 = run(prob, True, True, **kw)
    s0, p0 = run(prob, False, False, **kw)
    assert s1.num_iterations == s0.num_iterations
    assert abs(s1.final_cost - s0.final_cost) <= 1e-10 * s0.final_cost
    from oracle import oracle
    ref = prob.copy()
    st, s2 = oracle.solve(ref, abi.default_options(use_inner_iterations=0, max_num_iterations=6, **kw))
    assert st == 0 and abs(s1.final_cost - s2.final_cost) <= 1e-9 * s2.final_cost


def test_run_to_run_bit_identical():
    prob = synth.make_problem(60, 9000, 50000, seed=31, scene="ring", spread=0.4)
    kw = dict(point_dof=3, **IMPL)
    s1, p1 = run(prob, True, **kw)
    s2, p2 = run(prob, True, **kw)
    assert s1.final_cost == s2.final_cost
    assert np.array_equal(p1.extrinsics, p2.extrinsics) and np.array_equal(p1.points, p2.points)


def _setup_says(prob, capfd, **kw):
    saved = {k: os.environ.pop(k, None) for k in ENV}
    os.environ["TMI_BA_SETUP_TIMING"] = "1"
    os.environ["TMI_BA_MF_ONE_SWEEP"] = "1"
    try:
        capfd.readouterr()
        s = lib.Solver(prob.copy(), abi.default_options(max_num_iterations=1, **kw), 0, 1)
        s.close()
        err = capfd.readouterr().err
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    assert "camera side of matrix-free iterations" in err
    return "no camera-major records" in err


def test_the_records_stay_where_something_else_reads_them(capfd):
    prob = synth.make_problem(30, 3000, 15000, seed=29, scene="ring", spread=0.5)
    assert _setup_says(prob, capfd, point_dof=3, **IMPL)
    # the formed S gathers the records
    assert not _setup_says(prob, capfd, point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=abi.SCHUR_EXPLICIT)
    # fp32 evaluation keeps its own kernels
    assert not _setup_says(prob, capfd, point_dof=3, residual_precision=32, **IMPL)
    # the exact solvers form S
    assert not _setup_says(prob, capfd, point_dof=3, linear_solver_type=abi.DENSE_SCHUR)


def test_operator_info_reports_the_paths_of_a_handle():
    # tmi_ba_solver_operator_info: what bench.py prints as config.engine_paths
    prob = synth.make_problem(30, 3000, 15000, seed=29, scene="ring", spread=0.5)
    saved = {k: os.environ.pop(k, None) for k in ENV}
    os.environ["TMI_BA_MF_ONE_SWEEP"] = "1"
    try:
        s = lib.Solver(prob.copy(), abi.default_options(max_num_iterations=1, point_dof=3, **IMPL), 0, 1)
        info = s.operator_info()
        s.close()
        assert info["one_sweep_product"] and info["position_columns_formed"] and info["direct_camera_side"]
        assert info["implicit"] and not info["adaptive"]
        s = lib.Solver(prob.copy(), abi.default_options(max_num_iterations=1, point_dof=3, linear_solver_type=abi.DENSE_SCHUR), 0, 1)
        info = s.operator_info()
        s.close()
        assert not info["one_sweep_product"] and not info["direct_camera_side"] and not info["implicit"]
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
