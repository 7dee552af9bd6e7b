"""GPU: a problem several times the Venice size through create + a few LM iterations (headroom check:
32-bit index limits, memory, set-up time).  python tools/big_probe.py [cameras tracks observations [auto]]"""
import sys
import time

sys.path.insert(0, ".")
from theiasfm_amd import abi, lib, synth

nc, npt, nobs = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (4000, 3000000, 15000000)
t = time.time()
P = synth.make_problem(nc, npt, nobs, seed=7, scene="ring", spread=0.08, heavy_tail=0.002)
print("generated", P.num_cameras, P.num_points, P.num_observations, "in %.1f s" % (time.time() - t), flush=True)
modes = ((abi.SCHUR_EXPLICIT, "explicit"), (abi.SCHUR_IMPLICIT, "implicit"))
if len(sys.argv) >= 5 and sys.argv[4] == "auto":
    modes = ((abi.SCHUR_AUTO, "auto"),)
for mode, name in modes:
    o = abi.default_options(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, use_inner_iterations=0, max_num_iterations=5,
                            schur_mode=mode, function_tolerance=-1.0, gradient_tolerance=-1.0, parameter_tolerance=-1.0)
    t = time.time()
    s = lib.Solver(P.copy(), o)
    tc = time.time() - t
    st, sm = s.solve(o)
    print(name, "create %.3f s" % tc, "status", st, "its", sm.num_iterations, "pcg", sm.num_linear_solver_iterations,
          "solve %.4f s" % sm.solve_time_in_seconds, "ms/it %.3f" % (1e3 * sm.solve_time_in_seconds / max(sm.num_iterations, 1)),
          "cost %.6e -> %.6e" % (sm.initial_cost, sm.final_cost), "pairs", sm.num_schur_pairs, "blocks", sm.num_schur_blocks,
          sm.message.decode(), flush=True)
    s.close()
