import sys, time
sys.path.insert(0, ".")
from theiasfm_amd import abi, lib, synth
for shape in ((12000, 1200000, 6000000), (1778, 993923, 5001946)):
    P = synth.make_problem(*shape, seed=7, scene="ring", spread=0.08 if shape[0] > 2000 else 0.12, heavy_tail=0.002)
    for mode, name in ((abi.SCHUR_AUTO, "auto"), (abi.SCHUR_EXPLICIT, "explicit"), (abi.SCHUR_IMPLICIT, "implicit")):
        o = abi.default_options(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, use_inner_iterations=0, max_num_iterations=10,
                                schur_mode=mode, function_tolerance=-1.0, gradient_tolerance=-1.0, parameter_tolerance=-1.0)
        s = lib.Solver(P.copy(), o)
        s.solve(o); s.reset()
        st, sm = s.solve(o)
        print(shape[0], name, "ms/it %.3f" % (1e3 * sm.solve_time_in_seconds / sm.num_iterations), "pcg", sm.num_linear_solver_iterations,
              "matrix-free its", sm.num_matrix_free_iterations, "cost %.9e" % sm.final_cost, flush=True)
        s.close()
