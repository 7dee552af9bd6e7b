"""GPU: BASELINE config 5 at Venice size on one GPU: mixed camera models (50 % PINHOLE, 25 %
PINHOLE_RADIAL_TANGENTIAL, 25 % FISHEYE), intrinsics shared by groups of 8 views, fp32 residual
evaluation with fp64 accumulation; explicit operator (1 rank) and the matrix-free operator that
sharded runs use.  Prints ms per LM iteration and the parity of fp32 vs fp64 evaluation."""
import sys, time, json
sys.path.insert(0, ".")
import numpy as np
from theiasfm_amd import abi, lib, synth

bits = abi.INTRINSICS_FOCAL_LENGTH | abi.INTRINSICS_PRINCIPAL_POINTS | abi.INTRINSICS_RADIAL_DISTORTION | abi.INTRINSICS_TANGENTIAL_DISTORTION
P = synth.config("venice1778", models=[(abi.PINHOLE, 0.5), (abi.PINHOLE_RADIAL_TANGENTIAL, 0.25), (abi.FISHEYE, 0.25)],
                 shared_group_size=8, intrinsics_to_optimize=bits)
res = {}
for prec in (64, 32):
    for mode, name in ((abi.SCHUR_EXPLICIT, "explicit"), (abi.SCHUR_IMPLICIT, "implicit")):
        o = abi.default_options(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=mode,
                                residual_precision=prec, use_inner_iterations=0, max_num_iterations=8,
                                function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
        t0 = time.time()
        s = lib.Solver(P.copy(), o)
        t_create = time.time() - t0
        s.solve(abi.default_options(**{k: getattr(o, k) for k in ("point_dof", "linear_solver_type", "schur_mode", "residual_precision", "use_inner_iterations")}, max_num_iterations=2))
        s.reset()
        st, sm = s.solve(o)
        s.close()
        res[(prec, name)] = sm
        print(json.dumps(dict(precision=prec, operator=name, status=st, iterations=int(sm.num_iterations),
                              ms_per_iteration=round(1e3 * sm.solve_time_in_seconds / max(1, sm.num_iterations), 3),
                              pcg=int(sm.num_linear_solver_iterations), blocks=int(sm.num_reduced_blocks), D=int(sm.reduced_block_dim),
                              upper_blocks=int(sm.num_schur_blocks), pairs=int(sm.num_schur_pairs),
                              initial_rmse=round(sm.initial_rmse, 6), final_rmse=sm.final_rmse, final_cost=sm.final_cost,
                              create_s=round(t_create, 2))))
a, b = res[(64, "explicit")], res[(32, "explicit")]
print("fp32 vs fp64 final RMSE diff %.3e px, cost rel diff %.3e" % (abs(a.final_rmse - b.final_rmse), abs(a.final_cost - b.final_cost) / a.final_cost))
a, b = res[(64, "explicit")], res[(64, "implicit")]
print("implicit vs explicit (fp64) cost rel diff %.3e, PCG %d vs %d" % (abs(a.final_cost - b.final_cost) / a.final_cost, b.num_linear_solver_iterations, a.num_linear_solver_iterations))
