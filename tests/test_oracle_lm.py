"""The oracle's Ceres-semantics LM on the reference fixture and on synthetic
problems (CPU).  These are the pins the reference offers at this boundary
(SURVEY 8c): the fixture's known-answer cost / RMSE, "re-running BA on an
already adjusted reconstruction must not increase the cost", and exact vs
iterative Schur solvers agreeing on the minimum."""
import json
import os

import numpy as np

from oracle import oracle
from theiasfm_amd import abi, synth


def test_fountain11_known_answer(golden_dir):
    prob = abi.Problem.load(os.path.join(golden_dir, "fountain11_flat.npz"))
    known = json.load(open(os.path.join(golden_dir, "fountain11_known.json")))
    assert (prob.num_cameras, prob.num_groups, prob.num_points, prob.num_observations) == \
        (11, 1, 16616, 75022)
    cost, rmse, bad = oracle.cost(prob)
    assert bad == 0
    assert abs(cost - 7337.480164) < 5e-7      # SURVEY section 4 / BASELINE.md
    assert abs(rmse - 0.442277) < 5e-7
    assert abs(cost - known["cost"]) < 1e-9 * cost
    w = prob.points[:, 3]
    assert 0.9947 < w.min() < 0.9948 and 1.0081 < w.max() < 1.0082


def test_fountain11_ba_does_not_increase_cost(golden_dir):
    base = abi.Problem.load(os.path.join(golden_dir, "fountain11_flat.npz"))
    base.set_intrinsics_to_optimize(abi.INTRINSICS_DEFAULT)
    finals = []
    for solver, dof in ((abi.SPARSE_SCHUR, 4), (abi.ITERATIVE_SCHUR, 4), (abi.DENSE_SCHUR, 3)):
        p = base.copy()
        st, s = oracle.solve(p, abi.default_options(linear_solver_type=solver, point_dof=dof))
        assert st == 0 and s.success == 1
        assert abs(s.initial_cost - 7337.480164) < 5e-7
        assert s.final_cost <= s.initial_cost
        # stays in a small neighbourhood of the (already adjusted) input
        assert np.abs(p.extrinsics - base.extrinsics).max() < 1e-2
        assert abs(s.final_rmse - 0.442277) < 5e-3
        finals.append(s.final_cost)
    assert max(finals) - min(finals) < 1e-5 * finals[0]


def test_synthetic_recovers_noise_floor():
    prob = synth.config("ladybug49")
    for solver in (abi.DENSE_SCHUR, abi.ITERATIVE_SCHUR):
        p = prob.copy()
        st, s = oracle.solve(p, abi.default_options(linear_solver_type=solver, point_dof=3))
        assert st == 0 and s.success == 1 and s.termination == 0
        dof = 2 * p.num_observations - 9 * p.num_cameras - 3 * p.num_points
        expect = 0.5 * np.sqrt(2.0 * dof / (2.0 * p.num_observations))
        assert abs(s.final_rmse - expect) < 0.02
        assert s.final_cost < 1e-2 * s.initial_cost


def test_constant_blocks_are_untouched():
    prob = synth.make_problem(6, 60, 360, seed=11, scene="allsee")
    prob.camera_flags[0] = abi.CAMERA_POSITION_CONSTANT | abi.CAMERA_ORIENTATION_CONSTANT
    prob.camera_flags[1] = abi.CAMERA_POSITION_CONSTANT
    prob.camera_flags[2] = abi.CAMERA_ORIENTATION_CONSTANT
    prob.point_constant[:5] = 1
    prob.set_intrinsics_to_optimize(abi.INTRINSICS_FOCAL_LENGTH)
    before = prob.copy()
    st, s = oracle.solve(prob, abi.default_options(linear_solver_type=abi.DENSE_SCHUR, point_dof=4,
                                                   max_num_iterations=10))
    assert st == 0 and s.final_cost < s.initial_cost
    assert (prob.extrinsics[0] == before.extrinsics[0]).all()
    assert (prob.extrinsics[1, :3] == before.extrinsics[1, :3]).all()
    assert (prob.extrinsics[1, 3:] != before.extrinsics[1, 3:]).any()
    assert (prob.extrinsics[2, 3:] == before.extrinsics[2, 3:]).all()
    assert (prob.points[:5] == before.points[:5]).all()
    assert (prob.points[5:, :3] != before.points[5:, :3]).any()
    K0, K1 = before.intrinsics.reshape(-1, 7), prob.intrinsics.reshape(-1, 7)
    assert (K0[:, 1:] == K1[:, 1:]).all() and (K0[:, 0] != K1[:, 0]).all()


def test_invalid_start_point_fails_without_touching_inputs():
    prob = synth.make_problem(3, 10, 30, seed=5, scene="allsee")
    prob.points[0, :3] = prob.extrinsics[0, :3]  # a track on a camera centre
    before = prob.copy()
    st, s = oracle.solve(prob, abi.default_options())
    assert st == 6 and s.success == 0
    assert (prob.points == before.points).all()


def test_inner_iterations_config1_on_and_off():
    """SURVEY 8(d) config 1 with use_inner_iterations on and off.  The coordinate-descent sweep after
    each trust-region step can only lower the candidate's cost: after the same number of LM
    iterations the run with inner iterations is at least as low, the sweeps are counted, and they
    stop once their relative gain drops below 1e-3 (Ceres inner_iteration_tolerance)."""
    prob = synth.config("tiny")
    res = {}
    for inner in (0, 1):
        p = prob.copy()
        o = abi.default_options(point_dof=4, linear_solver_type=abi.DENSE_SCHUR, use_inner_iterations=inner,
                                max_num_iterations=3, function_tolerance=0.0, parameter_tolerance=0.0,
                                gradient_tolerance=0.0)
        st, s = oracle.solve(p, o)
        assert st == 0 and s.success == 1
        res[inner] = s
    assert res[0].num_inner_iteration_steps == 0
    assert 1 <= res[1].num_inner_iteration_steps <= 3
    assert res[1].final_cost <= res[0].final_cost * (1 + 1e-12)
    # run on: both settle in the same valley (all three cameras are free, so the problem keeps its
    # 7 gauge directions and the cost creeps for a long time: agreement to 1e-3, not to round-off)
    fin = {}
    for inner in (0, 1):
        p = prob.copy()
        st, s = oracle.solve(p, abi.default_options(point_dof=3, linear_solver_type=abi.DENSE_SCHUR,
                                                    use_inner_iterations=inner, max_num_iterations=50,
                                                    function_tolerance=1e-12))
        fin[inner] = s
    assert abs(fin[1].final_cost - fin[0].final_cost) <= 1e-3 * fin[0].final_cost
    assert fin[1].num_inner_iteration_steps < fin[1].num_iterations  # switched off before the end
