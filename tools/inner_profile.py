"""GPU: one solve of the Venice-sized heavy problem with inner iterations on (for rocprofv3 --kernel-trace)."""
import sys
sys.path.insert(0, ".")
from theiasfm_amd import abi, lib, synth

P = synth.config("venice1778_heavy")
o = abi.default_options(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, use_inner_iterations=1,
                        max_num_iterations=5, function_tolerance=-1.0, gradient_tolerance=-1.0, parameter_tolerance=-1.0)
s = lib.Solver(P, o)
st, sm = s.solve(o)
s.reset()
st, sm = s.solve(o)
print("its", sm.num_iterations, "inner steps", sm.num_inner_iteration_steps, "solve %.4f s" % sm.solve_time_in_seconds,
      "cost %.9e" % sm.final_cost)
s.close()
