"""-m gpu: the dense Cholesky of the exact reduced solve on its own (tools/chol_harness.hip wraps the kernels of
theiasfm_amd/csrc/dense_cholesky_df.h and dense_cholesky.h behind a C entry point) against numpy.

The dataflow kernel hands tiles between workgroups with flags and no fences (the hardware assumption is spelled out in
kernels.h), so beside accuracy this checks what a broken hand-over would break: every size class of the tile grid
(one tile, ragged last tile, exactly full tiles, more tiles than CUs), many repetitions, bit-identical results
whatever the timing, and agreement with the launch-per-panel path."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness():
    import __graft_entry__ as entry
    # torch first, as theiasfm_amd.lib does: the process must end up with ONE HIP runtime (torch ships its own
    # libamdhip64; a library that pulls in /opt/rocm's before torch initialises leaves torch without a device)
    import torch  # noqa: F401
    return C.CDLL(entry.build_chol_harness())


def run(fn, A, b, reps):
    n = A.shape[0]
    x = np.zeros(n)
    ms = C.c_double()
    info = (C.c_int * 8)()
    rc = fn(A.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), n, reps,
            C.byref(ms), info)
    assert rc == 0
    return x, ms.value, list(info)


def spd(n, seed):
    rng = np.random.default_rng(seed)
    M = rng.normal(size=(n, max(n // 2, 4)))
    A = M @ M.T + np.diag(rng.uniform(0.5, 2.0, n)) * n * 0.05
    d = np.exp(rng.normal(0, 1.0, n))
    return np.ascontiguousarray(A * d[:, None] * d[None, :]), rng.normal(size=n)


@pytest.mark.parametrize("n", [1, 5, 63, 64, 65, 127, 128, 129, 441, 1000, 1408, 1500, 3000])
def test_dataflow_cholesky_against_numpy(harness, n):
    A, b = spd(n, n)
    ref = np.linalg.solve(A, b)
    x, ms, info = run(harness.chol_df_solve, A, b, reps=8)
    assert info[0] == 0 and info[1] == 0        # no abort, no non-positive pivot
    cond = np.linalg.cond(A)
    assert np.abs(x - ref).max() <= 1e-13 * cond * np.abs(ref).max()
    assert np.linalg.norm(A @ x - b) <= 1e-12 * np.sqrt(n) * np.linalg.norm(A, 2) * np.linalg.norm(x)
    x2, _, _ = run(harness.chol_df_solve, A, b, reps=5)
    assert (x == x2).all()                      # fixed summation order: the timing of the hand-overs must not show
    xp, _, _ = run(harness.chol_panels_solve, A, b, reps=1)
    assert np.abs(x - xp).max() <= 1e-13 * cond * np.abs(ref).max()


def test_dataflow_cholesky_flags_an_indefinite_matrix(harness):
    A, b = spd(200, 3)
    A[150, 150] = -1.0
    _, _, info = run(harness.chol_df_solve, A, b, reps=1)
    assert info[0] == 0 and info[1] == 1


def test_engine_fallback_path_matches_the_dataflow_solve(monkeypatch):
    """The engine's fall-back for a dataflow launch that cannot become co-resident is the launch-per-panel Cholesky
    (engine.hip solve_reduced_dense); TMI_BA_CHOL_PANELS selects it outright: same LM trajectory as the default."""
    from theiasfm_amd import abi, lib, synth
    prob = synth.make_problem(120, 20000, 110000, seed=13, scene="ring", spread=0.4)  # n = 1080: 17 tile rows
    o = abi.default_options(linear_solver_type=abi.SPARSE_SCHUR, point_dof=3, max_num_iterations=5, use_inner_iterations=0,
                            function_tolerance=-1.0, gradient_tolerance=-1.0, parameter_tolerance=-1.0)
    a, b = prob.copy(), prob.copy()
    st_a, s_a = lib.solve(a, o)
    monkeypatch.setenv("TMI_BA_CHOL_PANELS", "1")
    st_b, s_b = lib.solve(b, o)
    assert st_a == st_b == 0 and s_a.num_iterations == s_b.num_iterations == 5
    assert s_a.num_successful_steps == s_b.num_successful_steps
    assert abs(s_a.final_cost - s_b.final_cost) <= 1e-10 * s_a.final_cost
    assert np.abs(a.extrinsics - b.extrinsics).max() <= 1e-8 * 100.0
