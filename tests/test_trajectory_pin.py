"""The TRAJECTORY of the trust-region layer against an independent model (tests/trajectory_model.py).

The oracle (CPU suite) and the device (-m gpu) are run with a per-iteration trace (tmi_ba_options.iteration_trace) and
held, iteration by iteration, to a Levenberg-Marquardt written from Ceres' documented rules on complex-step Jacobians
and the FULL normal equations (exact steps) or a dense Schur complement with Ceres' conjugate-gradient recurrences
(inexact steps): the same accepted / rejected / tolerance-reached sequence, the same trust-region radii, the same number
of PCG iterations in every LM iteration, costs and candidate costs to 1e-9 relative.

Cases: BASELINE config 1 (`tiny`, which has a rejected step), a robust loss (Huber, with gross outliers), a problem
with shared intrinsics groups, private intrinsics with both preconditioner block shapes, and BASELINE config 2's size
(ladybug49: 49 views / 7 776 tracks / 31 843 observations, 23 769 unknowns through SuperLU)."""
import numpy as np
import pytest

from oracle import oracle
from theiasfm_amd import abi, synth

import scipy_pin_model as M
import trajectory_model as T


def problem(name):
    if name == "tiny":
        return synth.config("tiny"), abi.LOSS_TRIVIAL, 2.0
    if name == "huber":
        prob, _, loss, width, _ = M.case("huber")
        return prob, loss, width
    if name == "cauchy":
        prob, _, loss, width, _ = M.case("cauchy")
        return prob, loss, width
    if name == "shared_group":
        return M.case("shared_group")[0], abi.LOSS_TRIVIAL, 2.0
    if name == "private":
        return M.small(), abi.LOSS_TRIVIAL, 2.0
    if name == "ladybug49":
        return synth.config("ladybug49"), abi.LOSS_TRIVIAL, 2.0
    raise ValueError(name)


# (problem, step, preconditioner block shape, LM iterations)
CASES = [
    ("tiny", "exact", "merged", 12),
    ("huber", "exact", "merged", 10),
    ("huber", "pcg", "merged", 10),
    ("huber", "pcg", "parameter_blocks", 10),
    ("cauchy", "pcg", "merged", 8),
    ("shared_group", "exact", "merged", 10),
    ("shared_group", "pcg", "merged", 10),
    ("private", "pcg", "parameter_blocks", 10),
    ("ladybug49", "exact", "merged", 8),
    ("ladybug49", "pcg", "merged", 8),
    ("ladybug49", "pcg", "parameter_blocks", 8),
]
_model_cache = {}


def model_rows(name, step, shape, iters):
    key = (name, step, shape, iters)
    if key not in _model_cache:
        prob, loss, width = problem(name)
        rows, _, why = T.levenberg_marquardt(T.Model(prob, 3, loss, width), solver=step, precond=shape, max_num_iterations=iters)
        _model_cache[key] = (rows, why)
    return _model_cache[key]


def options(name, step, shape, iters, exact_type=abi.DENSE_SCHUR):
    _, loss, width = problem(name)
    o = abi.default_options(
        point_dof=3, loss_function_type=loss, robust_loss_width=width, use_inner_iterations=0, max_num_iterations=iters,
        linear_solver_type=exact_type if step == "exact" else abi.ITERATIVE_SCHUR,
        preconditioner_type=abi.PRECOND_SCHUR_JACOBI if shape == "merged" else abi.PRECOND_SCHUR_JACOBI_PARAMETER_BLOCKS)
    return o, abi.attach_trace(o, iters + 1)


def hold(rows, why, trace, summary, pcg):
    n = summary.num_iterations
    msg = bytes(summary.message).split(b"\0")[0].decode()
    assert n == len(rows) and msg == why, (n, len(rows), msg, why)
    got = trace[:n]
    assert np.array_equal(got[:, 0], rows[:, 0]) and np.array_equal(got[:, 3], rows[:, 3]), (got[:, 3], rows[:, 3])
    assert np.abs(got[:, 1] - rows[:, 1]).max() <= 1e-9 * rows[0, 1]                  # cost at every linearisation point
    valid = rows[:, 3] >= 0
    assert np.all(np.abs(got[valid, 4] - rows[valid, 4]) <= 1e-9 * rows[valid, 1])    # candidate costs
    assert np.all(np.abs(got[:, 2] - rows[:, 2]) <= 1e-6 * rows[:, 2])                # trust-region radii
    assert np.all(np.abs(got[valid, 5] - rows[valid, 5]) <= 1e-6 * np.abs(rows[valid, 5]))  # model cost changes
    assert np.all(np.abs(got[valid, 7] - rows[valid, 7]) <= 1e-6 * rows[valid, 7])    # step norms
    if pcg:
        assert np.array_equal(got[:, 6], rows[:, 6]), (got[:, 6], rows[:, 6])        # the forcing sequence
        assert summary.num_linear_solver_iterations == int(rows[:, 6].sum())
    assert np.isnan(trace[n:]).all()


@pytest.mark.parametrize("name,step,shape,iters", CASES)
def test_oracle_follows_the_independent_trajectory(name, step, shape, iters):
    rows, why = model_rows(name, step, shape, iters)
    prob = problem(name)[0]
    o, trace = options(name, step, shape, iters)
    st, s = oracle.solve(prob.copy(), o)
    assert st == 0 and s.success == 1
    hold(rows, why, trace, s, step == "pcg")


def test_the_cases_exercise_the_rules():
    """the trajectories contain what they are meant to pin: a rejected step, growing and shrinking radii, termination by
    the function tolerance, PCG solves of different lengths"""
    rows, _ = model_rows("tiny", "exact", "merged", 12)
    assert (rows[:, 3] == 0).any() and (np.diff(rows[:, 2]) < 0).any() and (np.diff(rows[:, 2]) > 0).any()
    rows, why = model_rows("ladybug49", "pcg", "parameter_blocks", 8)
    assert why == "function tolerance reached" and rows[-1, 3] == 3 and len(set(rows[:, 6])) >= 3
    a, _ = model_rows("huber", "pcg", "merged", 10)
    b, _ = model_rows("huber", "pcg", "parameter_blocks", 10)
    assert a[:, 6].sum() < b[:, 6].sum()  # the block shape changes the forcing sequence, and the model sees it


def test_complex_step_jacobian_is_the_finite_difference_one():
    """the model's Jacobian against central differences (its only check that does not go through a trajectory)"""
    prob = M.small()
    m = T.Model(prob, 3)
    x = m.x0()
    J = m.jacobian(x).toarray()
    rng = np.random.default_rng(0)
    for j in rng.choice(m.n, 40, replace=False):
        h = 1e-6 * max(1.0, abs(x[j]))
        e = np.zeros(m.n)
        e[j] = h
        fd = (m.blocks(x + e) - m.blocks(x - e)).ravel() / (2 * h)
        assert np.abs(fd - J[:, j]).max() <= 1e-6 * max(1.0, np.abs(J[:, j]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("schur_mode", [abi.SCHUR_AUTO, abi.SCHUR_EXPLICIT, abi.SCHUR_IMPLICIT])
@pytest.mark.parametrize("name,step,shape,iters", CASES)
def test_device_follows_the_independent_trajectory(name, step, shape, iters, schur_mode):
    from theiasfm_amd import lib
    if step == "exact" and schur_mode != abi.SCHUR_AUTO:
        pytest.skip("schur_mode applies to ITERATIVE_SCHUR")
    rows, why = model_rows(name, step, shape, iters)
    prob = problem(name)[0]
    for exact_type in ((abi.DENSE_SCHUR, abi.SPARSE_SCHUR) if step == "exact" else (abi.DENSE_SCHUR,)):
        o, trace = options(name, step, shape, iters, exact_type)
        o.schur_mode = schur_mode
        st, s = lib.solve(prob.copy(), o)
        assert st == 0 and s.success == 1, bytes(s.message)
        hold(rows, why, trace, s, step == "pcg")
