"""-m gpu: handles are created and destroyed many times in a pipeline (BA -> filter -> select -> BA on
every estimator step): device memory in use must return to where it was."""
import pytest

from theiasfm_amd import abi, lib, synth

pytestmark = pytest.mark.gpu


def _used_mib():
    import torch
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20


def test_create_solve_side_kernels_destroy_does_not_leak_device_memory():
    P = synth.make_problem(60, 12000, 70000, seed=2, scene="ring", spread=0.3, heavy_tail=0.005)
    tv = synth.make_two_view_batch(200, 3, max_corr=80)
    base = None
    for it in range(5):
        for mode, solver in ((abi.SCHUR_EXPLICIT, abi.ITERATIVE_SCHUR), (abi.SCHUR_IMPLICIT, abi.ITERATIVE_SCHUR),
                             (abi.SCHUR_AUTO, abi.SPARSE_SCHUR)):
            o = abi.default_options(point_dof=3, linear_solver_type=solver, schur_mode=mode, max_num_iterations=3,
                                    use_inner_iterations=1)
            s = lib.Solver(P.copy(), o)
            st, _ = s.solve(o)
            assert st == 0
            s.filter_outlier_tracks(4.0, 2.0)
            s.select_good_tracks(10, 100, 100)
            s.adjust_tracks(abi.default_options(point_dof=3, max_num_iterations=10))
            s.close()
        lib.adjust_two_views(tv.copy(), 4)
        if it == 1:
            base = _used_mib()  # after the first rounds: module load, allocator pools
    assert _used_mib() - base < 8.0


def test_cluster_preconditioner_handles_do_not_leak_device_memory():
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_cluster_jacobi import options, shared_problem
    P = shared_problem(n_views=40, groups=(2, 12), seed=4)
    base = None
    for it in range(6):
        for mode in (abi.SCHUR_EXPLICIT, abi.SCHUR_AUTO):
            o = options(abi.PRECOND_CLUSTER_JACOBI, schur_mode=mode, max_num_iterations=3)
            s = lib.Solver(P.copy(), o)
            st, _ = s.solve(o)
            assert st == 0
            s.close()
        if it == 1:
            base = _used_mib()
    assert _used_mib() - base < 8.0


@pytest.mark.parametrize("solver,mode", [(abi.ITERATIVE_SCHUR, abi.SCHUR_AUTO), (abi.SPARSE_SCHUR, abi.SCHUR_AUTO)])
def test_set_parameters_on_a_resident_handle(solver, mode):
    """tmi_ba_solver_set_parameters: new parameter values for the resident structure -- what the shim's resident session
    does between two BundleAdjustReconstruction calls.  Solving from the uploaded values equals a fresh handle created
    with them, bit for bit, and reset() goes back to the uploaded values, not to the ones of create."""
    import numpy as np
    P = synth.make_problem(20, 2500, 12000, seed=6, scene="ring", spread=0.4)
    o = abi.default_options(point_dof=3, linear_solver_type=solver, schur_mode=mode, max_num_iterations=4,
                            use_inner_iterations=0)
    Q = P.copy()
    rng = np.random.default_rng(1)
    Q.extrinsics[:, :3] += 0.02 * rng.normal(size=(Q.num_cameras, 3))
    Q.points[:, :3] += 0.02 * rng.normal(size=(Q.num_points, 3))
    Q.intrinsics[0::7] *= 1.001
    fresh = lib.Solver(Q.copy(), o)
    st_f, s_f = fresh.solve(o)
    out_f = fresh.download().copy()
    fresh.close()
    s = lib.Solver(P.copy(), o)
    st0, s0 = s.solve(o)                      # something else happened on the handle before
    s.set_parameters(Q.copy())
    st1, s1 = s.solve(o)
    out1 = s.download().copy()
    s.reset()
    st2, s2 = s.solve(o)
    s.close()
    assert st_f == st0 == st1 == st2 == 0
    assert s1.initial_cost == s_f.initial_cost and s1.final_cost == s_f.final_cost
    assert (out1.extrinsics == out_f.extrinsics).all() and (out1.points == out_f.points).all()
    assert (out1.intrinsics == out_f.intrinsics).all()
    assert s2.initial_cost == s_f.initial_cost and s2.final_cost == s_f.final_cost
    assert s0.initial_cost != s1.initial_cost
