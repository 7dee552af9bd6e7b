"""PCG iterations per LM iteration of one problem through every path of the device and the oracle (how sensitive the
stopping iteration of a long PCG solve is to the summation order).  usage: python tools/pcg_count_probe.py"""
import os
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
from oracle import oracle  # noqa: E402
from theiasfm_amd import abi, lib, synth  # noqa: E402

prob = synth.make_problem(300, 60000, 300000, seed=5, scene="street", spread=0.025, heavy_tail=0.002)
iters = 6


def dev(mode, env):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        o = abi.default_options(use_inner_iterations=0, max_num_iterations=iters, linear_solver_type=abi.ITERATIVE_SCHUR,
                                schur_mode=mode, point_dof=3)
        tr = abi.attach_trace(o, iters + 1)
        st, s = lib.solve(prob.copy(), o)
        return tr[:int(s.num_iterations), 6].astype(int).tolist(), s.final_cost
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


o = abi.default_options(use_inner_iterations=0, max_num_iterations=iters, linear_solver_type=abi.ITERATIVE_SCHUR, point_dof=3)
tr = abi.attach_trace(o, iters + 1)
st, s = oracle.solve(prob.copy(), o)
print("oracle                         ", tr[:int(s.num_iterations), 6].astype(int).tolist(), s.final_cost)
print("explicit, persistent PCG       ", *dev(abi.SCHUR_EXPLICIT, {"TMI_BA_PCG_PERSISTENT": "1"}))
print("explicit, launch per step      ", *dev(abi.SCHUR_EXPLICIT, {"TMI_BA_PCG_PERSISTENT": "0"}))
print("implicit one sweep             ", *dev(abi.SCHUR_IMPLICIT, {"TMI_BA_MF_ONE_SWEEP": "1"}))
print("implicit two pass              ", *dev(abi.SCHUR_IMPLICIT, {"TMI_BA_MF_ONE_SWEEP": "0"}))
