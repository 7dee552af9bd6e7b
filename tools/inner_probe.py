"""GPU: inner iterations at scale: Venice-sized timing (device only), Alamo-sized parity vs the oracle."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from oracle import oracle
from theiasfm_amd import abi, lib, synth

P = synth.config("venice1778")
for inner in (0, 1):
    o = abi.default_options(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, use_inner_iterations=inner,
                            max_num_iterations=6, profile_kernels=1)
    s = lib.Solver(P.copy(), o)
    st, sm = s.solve(o)
    s.close()
    d = sm.as_dict()
    print("venice inner", inner, "its", sm.num_iterations, "inner steps", sm.num_inner_iteration_steps,
          "cost %.6e -> %.9e" % (sm.initial_cost, sm.final_cost), "solve %.3f s" % sm.solve_time_in_seconds,
          "linearize-class ms %.1f" % (1e3 * d["kernel_seconds"][0]), sm.message.decode())
A = synth.config("alamo")
o = abi.default_options(point_dof=3, linear_solver_type=abi.SPARSE_SCHUR, use_inner_iterations=1, max_num_iterations=4)
a, b = A.copy(), A.copy()
st_d, s_d = lib.solve(a, o)
t = time.time()
st_o, s_o = oracle.solve(b, o)
print("alamo device %d its cost %.12e | oracle %d its cost %.12e (%.1f s) rel diff %.2e inner steps %d/%d" % (
    s_d.num_iterations, s_d.final_cost, s_o.num_iterations, s_o.final_cost, time.time() - t,
    abs(s_d.final_cost - s_o.final_cost) / s_o.final_cost, s_d.num_inner_iteration_steps, s_o.num_inner_iteration_steps))
print("max |d ext|", np.abs(a.extrinsics - b.extrinsics).max(), "max |d pts|", np.abs(a.points - b.points).max())
