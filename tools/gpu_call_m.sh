#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python - <<'PY' 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -30
import sys
sys.path.insert(0,'.')
from theiasfm_amd import abi, lib, synth
P = synth.config("venice1778_heavy")
o = abi.default_options(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, use_inner_iterations=0, max_num_iterations=12, verbose=1,
                        function_tolerance=-1.0, gradient_tolerance=-1.0, parameter_tolerance=-1.0)
s = lib.Solver(P, o)
st, sm = s.solve(o)
print("pcg total", sm.num_linear_solver_iterations, "its", sm.num_iterations)
s.close()
PY
