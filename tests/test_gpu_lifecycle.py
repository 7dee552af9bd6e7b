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
