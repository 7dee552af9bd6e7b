"""-m gpu: randomised shapes for the record-free camera side and the view-by-view trial cost (direct_diag.h) against the
record path -- chunk boundaries (views with 1 ... several thousand observations, i.e. less than one trip up to several
chunks), views nobody observes, constant points / positions / orientations / cameras, every loss, both point
parameterisations, block widths 6 / 9 / 12, mixed camera models.  Same trajectory to round-off (iteration trace)."""
import os

import numpy as np
import pytest

from theiasfm_amd import abi, lib, synth

pytestmark = pytest.mark.gpu

ENV = ("TMI_BA_DIRECT_DIAG", "TMI_BA_MF_ONE_SWEEP", "TMI_BA_COST_BY_VIEW")
LOSSES = [abi.LOSS_TRIVIAL, abi.LOSS_HUBER, abi.LOSS_SOFTLONE, abi.LOSS_CAUCHY, abi.LOSS_ARCTAN, abi.LOSS_TUKEY]
INTR = [abi.INTRINSICS_NONE, abi.INTRINSICS_FOCAL_LENGTH, abi.INTRINSICS_DEFAULT,
        abi.INTRINSICS_FOCAL_LENGTH | abi.INTRINSICS_PRINCIPAL_POINTS,
        abi.INTRINSICS_FOCAL_LENGTH | abi.INTRINSICS_PRINCIPAL_POINTS | abi.INTRINSICS_RADIAL_DISTORTION]
MODELS = [abi.PINHOLE, abi.PINHOLE_RADIAL_TANGENTIAL, abi.FISHEYE, abi.FOV, abi.DIVISION_UNDISTORTION]


def solve(prob, direct, iters, **kw):
    saved = {k: os.environ.pop(k, None) for k in ENV}
    try:
        if not direct:
            os.environ["TMI_BA_DIRECT_DIAG"] = "0"
        os.environ["TMI_BA_MF_ONE_SWEEP"] = "1"
        p = prob.copy()
        o = abi.default_options(use_inner_iterations=0, max_num_iterations=iters, linear_solver_type=abi.ITERATIVE_SCHUR,
                                schur_mode=abi.SCHUR_IMPLICIT, **kw)
        tr = abi.attach_trace(o, iters + 1)
        st, s = lib.solve(p, o)
        return st, s, p, tr
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


@pytest.mark.parametrize("seed", list(range(40)))
def test_random_shapes(seed):
    rng = np.random.default_rng(1000 + seed)
    n_cam = int(rng.choice([7, 12, 20, 45, 90, 150]))  # (three views with two observations per track are a
    # degenerate geometry: PCG runs 25 iterations on 27 unknowns and its length is decided by round-off)
    n_pts = int(rng.integers(40 * n_cam, 160 * n_cam))
    k = float(rng.uniform(2.5, min(8.0, 0.8 * n_cam)))
    n_obs = int(k * n_pts)
    models = None
    if rng.random() < 0.5:
        pick = rng.choice(len(MODELS), size=int(rng.integers(1, 4)), replace=False)
        models = [(MODELS[int(i)], 1.0 / len(pick)) for i in pick]
    prob = synth.make_problem(n_cam, n_pts, n_obs, seed=2000 + seed, scene="ring", spread=float(rng.uniform(0.3, 1.0)),
                              models=models, intrinsics_to_optimize=int(rng.choice(INTR)),
                              heavy_tail=float(rng.choice([0.0, 0.0, 0.01])))
    if rng.random() < 0.4:
        prob.point_constant[rng.choice(prob.num_points, size=max(1, prob.num_points // 50), replace=False)] = 1
    if rng.random() < 0.4:
        cams = rng.choice(n_cam, size=max(1, n_cam // 6), replace=False)
        for c in cams:
            prob.camera_flags[c] = int(rng.choice([abi.CAMERA_POSITION_CONSTANT, abi.CAMERA_ORIENTATION_CONSTANT,
                                                   abi.CAMERA_POSITION_CONSTANT | abi.CAMERA_ORIENTATION_CONSTANT]))
    kw = dict(point_dof=int(rng.choice([3, 4])), loss_function_type=int(rng.choice(LOSSES)),
              robust_loss_width=float(rng.uniform(1.0, 4.0)), jacobi_scaling=int(rng.random() < 0.85))
    if rng.random() < 0.3:
        kw["preconditioner_type"] = abi.PRECOND_SCHUR_JACOBI_PARAMETER_BLOCKS
    st1, s1, p1, t1 = solve(prob, True, 5, **kw)
    st0, s0, p0, t0 = solve(prob, False, 5, **kw)
    assert st1 == st0, (st1, s1.message, st0, s0.message)
    if st0 != 0:
        return  # (a shape the engine refuses: both paths must refuse it alike)
    n = int(s0.num_iterations)
    assert s1.num_iterations == s0.num_iterations
    assert np.array_equal(t1[:n, 3], t0[:n, 3]), (t1[:n, 3], t0[:n, 3])  # accepted / rejected / invalid
    assert np.array_equal(t1[:n, 6], t0[:n, 6]), (t1[:n, 6], t0[:n, 6])  # PCG iterations of every LM iteration
    assert np.abs(t1[:n, 1] - t0[:n, 1]).max() <= 1e-9 * max(t0[0, 1], 1e-30)
    assert abs(s1.final_cost - s0.final_cost) <= 1e-9 * max(s0.final_cost, 1e-30)


@pytest.mark.parametrize("seed", list(range(16)))
def test_random_shapes_follow_the_oracle(seed):
    """The same generator against the ORACLE (default paths of the engine, whatever they are for the shape): exact and
    iterative reduced solves, row by row of the iteration trace."""
    from oracle import oracle
    rng = np.random.default_rng(5000 + seed)
    n_cam = int(rng.choice([7, 12, 20, 45]))
    n_pts = int(rng.integers(40 * n_cam, 120 * n_cam))
    n_obs = int(float(rng.uniform(2.5, min(7.0, 0.8 * n_cam))) * n_pts)
    models = None
    if rng.random() < 0.5:
        pick = rng.choice(len(MODELS), size=int(rng.integers(1, 4)), replace=False)
        models = [(MODELS[int(i)], 1.0 / len(pick)) for i in pick]
    prob = synth.make_problem(n_cam, n_pts, n_obs, seed=6000 + seed, scene="ring", spread=float(rng.uniform(0.3, 1.0)),
                              models=models, intrinsics_to_optimize=int(rng.choice(INTR)))
    if rng.random() < 0.4:
        prob.point_constant[rng.choice(prob.num_points, size=max(1, prob.num_points // 50), replace=False)] = 1
    if rng.random() < 0.4:
        for c in rng.choice(n_cam, size=max(1, n_cam // 6), replace=False):
            prob.camera_flags[c] = int(rng.choice([abi.CAMERA_POSITION_CONSTANT, abi.CAMERA_ORIENTATION_CONSTANT]))
    solver = int(rng.choice([abi.DENSE_SCHUR, abi.SPARSE_SCHUR, abi.ITERATIVE_SCHUR]))
    kw = dict(point_dof=int(rng.choice([3, 4])), loss_function_type=int(rng.choice(LOSSES)),
              robust_loss_width=float(rng.uniform(1.0, 4.0)), linear_solver_type=solver, use_inner_iterations=0,
              max_num_iterations=5)
    saved = {k: os.environ.pop(k, None) for k in ENV}
    try:
        os.environ["TMI_BA_MF_ONE_SWEEP"] = "1"
        od = abi.default_options(**kw)
        td = abi.attach_trace(od, 6)
        a = prob.copy()
        st_d, s_d = lib.solve(a, od)
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    oo = abi.default_options(**kw)
    to = abi.attach_trace(oo, 6)
    b = prob.copy()
    st_o, s_o = oracle.solve(b, oo)
    assert st_d == st_o, (st_d, s_d.message, st_o, s_o.message)
    if st_o != 0:
        return
    n = int(s_o.num_iterations)
    assert s_d.num_iterations == s_o.num_iterations
    assert np.array_equal(td[:n, 3], to[:n, 3]), (td[:n, 3], to[:n, 3])
    assert np.array_equal(td[:n, 6], to[:n, 6]), (td[:n, 6], to[:n, 6])
    tol = 1e-9 if kw["point_dof"] == 3 else 1e-7  # (4-dof points carry an exact gauge direction)
    assert np.abs(td[:n, 1] - to[:n, 1]).max() <= tol * max(to[0, 1], 1e-30)
    assert abs(s_d.final_cost - s_o.final_cost) <= tol * max(s_o.final_cost, 1e-30)
