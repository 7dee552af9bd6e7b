"""GPU: create / solve / side kernels / destroy in a loop; device memory in use must not grow."""
import sys
sys.path.insert(0, ".")
import torch
from theiasfm_amd import abi, lib, synth

P = synth.config("alamo")
tv = synth.make_two_view_batch(500, 3, max_corr=100)


def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20


base = None
for it in range(12):
    for mode, solver in ((abi.SCHUR_EXPLICIT, abi.ITERATIVE_SCHUR), (abi.SCHUR_IMPLICIT, abi.ITERATIVE_SCHUR), (abi.SCHUR_AUTO, abi.SPARSE_SCHUR)):
        o = abi.default_options(point_dof=3, linear_solver_type=solver, schur_mode=mode, max_num_iterations=3, use_inner_iterations=1)
        s = lib.Solver(P.copy(), o)
        s.solve(o)
        s.filter_outlier_tracks(4.0, 2.0)
        s.select_good_tracks(10, 100, 100)
        s.adjust_tracks(abi.default_options(point_dof=3, max_num_iterations=10))
        s.close()
    lib.adjust_two_views(tv.copy(), 4)
    u = used()
    if it == 1:
        base = u
    print("round", it, "device MiB in use %.1f" % u, flush=True)
print("growth after warm-up: %.1f MiB" % (used() - base))
