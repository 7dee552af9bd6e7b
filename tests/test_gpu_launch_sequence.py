"""-m gpu: the round-6 launch sequence of an LM iteration against the one it replaces.

Round 6 cut the launches of an LM iteration (engine.hip, tmi_ba_solver_solve -- the replacement of ceres::Solve at
src/theia/sfm/bundle_adjustment/bundle_adjuster.cc:205) from ~36 to ~27:

* `ddg::camera_finish_kernel` = camera_diag_direct_reduce + finish_diag + precond_invert + pcg_init in one launch
  (TMI_BA_FUSED_FINISH=0 keeps the separate launches),
* PCG's next iteration is enqueued before the host has read the current one's stopping test, its kernels return at once
  when the test held (TMI_BA_PCG_SPECULATE=0: the host reads first),
* the start of a solve writes no planes before their Jacobi scales are known: a norms-only `linearize` (cost + the point
  columns' scales) and `camera_diag_direct` on records that carry the points alone replace linearize + point_scale +
  point_eliminate (TMI_BA_FAST_START=0: the three-pass start),
* `pcg_step`'s last workgroup forms p (was `pcg_p`), `back_substitute` writes the candidate points (was `update_points`),
  `linearize` leaves the position coefficients (was `pos_coef`), `update_cameras` the scaled copy of y_c (was `pos_scale`).

The fused kernel sums rho in pcg_init's order, so the switches must give the SAME BITS: iteration trace, PCG counts and final
parameters -- over the matrix-free and the formed operator, the SCHUR_JACOBI shapes, IDENTITY, robust losses, both point
parameterisations, shared intrinsics blocks, a solve that stops on its PCG iteration limit and one whose steps are
rejected."""
import os

import numpy as np
import pytest

from theiasfm_amd import abi, lib, synth

pytestmark = pytest.mark.gpu

ENV = ("TMI_BA_FUSED_FINISH", "TMI_BA_PCG_SPECULATE", "TMI_BA_MF_ONE_SWEEP", "TMI_BA_ATTACH_EVENTS", "TMI_BA_FAST_START")


def run(prob, fused, speculate, attach=True, profile=0, fast_start=True, **kw):
    saved = {k: os.environ.pop(k, None) for k in ENV}
    try:
        os.environ["TMI_BA_FUSED_FINISH"] = "1" if fused else "0"
        os.environ["TMI_BA_PCG_SPECULATE"] = "1" if speculate else "0"
        os.environ["TMI_BA_ATTACH_EVENTS"] = "1" if attach else "0"
        os.environ["TMI_BA_FAST_START"] = "1" if fast_start else "0"
        os.environ["TMI_BA_MF_ONE_SWEEP"] = "1"  # (the one-sweep product below its size threshold: its kernels carry the guard)
        p = prob.copy()
        kw.setdefault("max_num_iterations", 8)
        o = abi.default_options(use_inner_iterations=kw.pop("use_inner_iterations", 0), profile_kernels=profile, **kw)
        trace = abi.attach_trace(o, kw["max_num_iterations"])
        st, s = lib.solve(p, o)
        assert st == 0, s.message
        return s, p, trace
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def same_bits(a, b):
    sa, pa, ta = a
    sb, pb, tb = b
    assert sa.num_iterations == sb.num_iterations and sa.num_linear_solver_iterations == sb.num_linear_solver_iterations
    assert sa.num_successful_steps == sb.num_successful_steps
    assert sa.final_cost == sb.final_cost
    assert np.array_equal(ta, tb, equal_nan=True)
    assert np.array_equal(pa.points, pb.points) and np.array_equal(pa.extrinsics, pb.extrinsics)
    assert np.array_equal(pa.intrinsics, pb.intrinsics)


IMPL = dict(linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=abi.SCHUR_IMPLICIT)
EXPL = dict(linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=abi.SCHUR_EXPLICIT)


CASES = {
    "matrix_free_dof3": (lambda: synth.make_problem(60, 9000, 50000, seed=41, scene="ring", spread=0.4), dict(point_dof=3, **IMPL)),
    "matrix_free_dof4_parameter_blocks": (lambda: synth.make_problem(60, 9000, 50000, seed=43, scene="ring", spread=0.4),
                                          dict(point_dof=4, preconditioner_type=abi.PRECOND_SCHUR_JACOBI_PARAMETER_BLOCKS, **IMPL)),
    "matrix_free_identity_cauchy": (lambda: synth.make_problem(40, 6000, 32000, seed=45, scene="ring", spread=0.5, heavy_tail=0.01),
                                    dict(point_dof=3, preconditioner_type=abi.PRECOND_IDENTITY, loss_function_type=abi.LOSS_CAUCHY,
                                         robust_loss_width=3.0, **IMPL)),
    "formed_S": (lambda: synth.make_problem(50, 7000, 40000, seed=47, scene="ring", spread=0.5), dict(point_dof=3, **EXPL)),
    "auto": (lambda: synth.make_problem(50, 7000, 40000, seed=49, scene="ring", spread=0.5),
             dict(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=abi.SCHUR_AUTO)),
    "pcg_iteration_limit": (lambda: synth.make_problem(60, 9000, 50000, seed=51, scene="ring", spread=0.4),
                            dict(point_dof=3, max_linear_solver_iterations=3, **IMPL)),
    "residual_reset_iterations": (lambda: synth.make_problem(80, 9000, 50000, seed=53, scene="street", spread=0.4),
                                  dict(point_dof=3, preconditioner_type=abi.PRECOND_IDENTITY, eta=1e-6, max_linear_solver_iterations=35,
                                       max_num_iterations=3, **IMPL)),
    "rejected_steps": (lambda: synth.make_problem(40, 5000, 28000, seed=55, scene="ring", spread=0.5),
                       dict(point_dof=3, initial_trust_region_radius=1e12, max_num_iterations=10, **IMPL)),
    "inner_iterations": (lambda: synth.make_problem(40, 5000, 28000, seed=57, scene="ring", spread=0.5),
                         dict(point_dof=4, use_inner_iterations=1, max_num_iterations=5, **IMPL)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_fused_finish_and_speculative_pcg_give_the_same_bits(name):
    make, kw = CASES[name]
    prob = make()
    ref = run(prob, fused=False, speculate=False, **dict(kw))
    assert ref[0].num_linear_solver_iterations > 0
    for fused, spec in ((True, False), (False, True), (True, True)):
        same_bits(ref, run(prob, fused=fused, speculate=spec, **dict(kw)))


def test_shared_intrinsics_blocks_take_the_fused_finish():
    """shared blocks ride behind the view blocks: the fused launch covers them (no direct diagonal there: its non-direct form)"""
    prob = synth.make_problem(36, 5000, 30000, seed=61, scene="ring", spread=0.5, shared_group_size=12)
    kw = dict(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=abi.SCHUR_EXPLICIT)
    same_bits(run(prob, False, False, **dict(kw)), run(prob, True, True, **dict(kw)))


def test_voided_speculation_is_not_counted():
    """a speculative PCG iteration that found the solve stopped did nothing: the launch counts and the timed launches of
    the product class are those of the run without speculation, and the events handed to the launches
    (hipExtLaunchKernelGGL) time what the recorded ones time"""
    prob = synth.make_problem(60, 9000, 50000, seed=41, scene="ring", spread=0.4)
    kw = dict(point_dof=3, **IMPL)
    a = run(prob, True, False, attach=False, profile=1, **dict(kw))[0]
    b = run(prob, True, True, attach=True, profile=1, **dict(kw))[0]
    spmv = abi.KERNEL_CLASS_NAMES.index("spmv")
    vec = abi.KERNEL_CLASS_NAMES.index("pcg_vector")
    assert a.kernel_launches[spmv] == b.kernel_launches[spmv] == a.num_linear_solver_iterations
    assert a.kernel_launches[vec] == b.kernel_launches[vec]
    assert a.kernel_seconds[spmv] > 0 and b.kernel_seconds[spmv] > 0
    # the attached events leave out the two barrier packets: never longer than the recorded ones by more than noise
    assert b.kernel_seconds[spmv] < 1.5 * a.kernel_seconds[spmv] + 1e-4
    assert b.kernel_seconds[spmv] > 0.3 * a.kernel_seconds[spmv]


@pytest.mark.parametrize("name", ["matrix_free_dof3", "matrix_free_dof4_parameter_blocks", "matrix_free_identity_cauchy", "auto",
                                  "rejected_steps"])
def test_fast_start_gives_the_same_bits(name):
    """the norms-only first pass sums what point_scale summed, in its order; the U diagonal comes out of the same kernel:
    scale_p, scale_c and with them the whole trajectory are the three-pass start's, bit for bit"""
    make, kw = CASES[name]
    prob = make()
    same_bits(run(prob, True, True, fast_start=False, **dict(kw)), run(prob, True, True, fast_start=True, **dict(kw)))


def test_fast_start_mixed_models_and_constant_blocks():
    """the generic body of the norms-only pass (camera-model switch, column masks), constant points, a constant position"""
    prob = synth.make_problem(48, 6000, 36000, seed=33, scene="ring", spread=0.5,
                              models=[(abi.PINHOLE, 0.2), (abi.PINHOLE_RADIAL_TANGENTIAL, 0.2), (abi.FISHEYE, 0.2), (abi.FOV, 0.2),
                                      (abi.DIVISION_UNDISTORTION, 0.2)],
                              intrinsics_to_optimize=abi.INTRINSICS_FOCAL_LENGTH | abi.INTRINSICS_PRINCIPAL_POINTS)
    prob.point_constant[5] = 1
    prob.point_constant[77] = 1
    prob.camera_flags[3] = abi.CAMERA_ORIENTATION_CONSTANT
    kw = dict(point_dof=3, loss_function_type=abi.LOSS_SOFTLONE, robust_loss_width=2.0, **IMPL)
    same_bits(run(prob, True, True, fast_start=False, **dict(kw)), run(prob, True, True, fast_start=True, **dict(kw)))
