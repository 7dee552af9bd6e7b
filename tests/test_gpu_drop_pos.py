"""-m gpu: the position columns of the camera Jacobian formed from the point block (DeviceView::drop_pos, round 4)
against the stored columns (TMI_BA_DROP_POS=0).

reprojection_error.h:60-95: the residual depends on the camera position C and the homogeneous point X only through
X[0..2] - X[3] C, so d r / d C = -X[3] d r / d X[0..2] column by column; the Jacobi scales and the loss corrector are
linear maps applied to both.  The engine therefore does not store the three position columns of the A planes where
every camera block has its position free and no point is constant, and linearize / point_eliminate / back_substitute /
the one-sweep product form them from the Jp planes.  Same trajectory to round-off, whatever the solver type, the loss,
the point parameterisation and the block width; and the switch really is off where the identity does not hold."""
import os

import numpy as np
import pytest

from theiasfm_amd import abi, lib, synth

pytestmark = pytest.mark.gpu


def run(prob, drop, one_sweep, **kw):
    saved = {k: os.environ.pop(k, None) for k in ("TMI_BA_DROP_POS", "TMI_BA_MF_ONE_SWEEP", "TMI_BA_SETUP_TIMING")}
    try:
        if not drop:
            os.environ["TMI_BA_DROP_POS"] = "0"
        if one_sweep:
            os.environ["TMI_BA_MF_ONE_SWEEP"] = "1"
        p = prob.copy()
        st, s = lib.solve(p, abi.default_options(max_num_iterations=6, use_inner_iterations=0, **kw))
        assert st == 0, s.message
        return s, p
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


CASES = {
    # the exact solver: only linearize / point_eliminate / back_substitute read the planes
    "dense_schur": (lambda: synth.config("ladybug49"), False, dict(point_dof=3, linear_solver_type=abi.DENSE_SCHUR)),
    # PCG on the formed S, Huber, homogeneous points with four free coordinates
    "explicit_huber_dof4": (lambda: synth.make_problem(40, 6000, 30000, seed=21, scene="ring", spread=0.5), False,
                            dict(point_dof=4, linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=abi.SCHUR_EXPLICIT,
                                 loss_function_type=abi.LOSS_HUBER, robust_loss_width=2.0)),
    # the one-sweep product, Cauchy, tracks of up to 300 views
    "one_sweep_cauchy": (lambda: synth.make_problem(320, 20000, 120000, seed=23, scene="ring", spread=0.6, heavy_tail=0.01), True,
                         dict(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=abi.SCHUR_IMPLICIT,
                              loss_function_type=abi.LOSS_CAUCHY, robust_loss_width=3.0)),
    # extrinsics only: the block is [position | rotation]
    "one_sweep_d6": (lambda: synth.make_problem(60, 9000, 50000, seed=25, scene="ring", spread=0.4,
                                                intrinsics_to_optimize=abi.INTRINSICS_NONE), True,
                     dict(point_dof=4, linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=abi.SCHUR_IMPLICIT)),
    # 16-wide blocks
    "one_sweep_d16": (lambda: synth.make_problem(30, 4000, 22000, seed=27, scene="ring", spread=0.5,
                                                 models=[(abi.PINHOLE_RADIAL_TANGENTIAL, 1.0)],
                                                 intrinsics_to_optimize=abi.INTRINSICS_ALL & ~(abi.INTRINSICS_SKEW | abi.INTRINSICS_ASPECT_RATIO)), True,
                      dict(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=abi.SCHUR_IMPLICIT)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_formed_position_columns_give_the_stored_ones_trajectory(name):
    make, one_sweep, kw = CASES[name]
    prob = make()
    s1, p1 = run(prob, True, one_sweep, **kw)
    s0, p0 = run(prob, False, one_sweep, **kw)
    assert s1.num_iterations == s0.num_iterations and s1.num_successful_steps == s0.num_successful_steps
    assert s1.num_linear_solver_iterations == s0.num_linear_solver_iterations
    assert abs(s1.final_cost - s0.final_cost) <= 1e-11 * s0.final_cost
    scale = max(1.0, np.abs(p0.extrinsics).max())
    assert np.abs(p1.extrinsics - p0.extrinsics).max() <= 1e-9 * scale
    assert np.abs(p1.intrinsics - p0.intrinsics).max() <= 1e-9 * max(1.0, np.abs(p0.intrinsics).max())


def _says_formed(prob, capfd, **kw):
    os.environ["TMI_BA_SETUP_TIMING"] = "1"
    try:
        capfd.readouterr()
        s = lib.Solver(prob.copy(), abi.default_options(max_num_iterations=1, **kw), 0, 1)
        s.close()
        err = capfd.readouterr().err
    finally:
        os.environ.pop("TMI_BA_SETUP_TIMING", None)
    assert "position columns of the A planes" in err
    return "formed from Jp" in err


def test_the_columns_stay_stored_where_the_identity_does_not_hold(capfd):
    kw = dict(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=abi.SCHUR_EXPLICIT)
    prob = synth.make_problem(30, 3000, 15000, seed=29, scene="ring", spread=0.5)
    assert _says_formed(prob, capfd, **kw)
    const_pt = prob.copy()
    const_pt.point_constant[5] = 1  # its Jp block is stored as zero
    assert not _says_formed(const_pt, capfd, **kw)
    const_pos = prob.copy()
    const_pos.camera_flags[3] = abi.CAMERA_POSITION_CONSTANT  # its block starts with the rotation columns
    assert not _says_formed(const_pos, capfd, **kw)
    assert not _says_formed(prob, capfd, **dict(kw, residual_precision=32))
    # the two-pass matrix-free kernels read stored columns (below the size from which the one-sweep product is built)
    assert not _says_formed(prob, capfd, **dict(kw, schur_mode=abi.SCHUR_IMPLICIT))
