#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -8
python tools/inner_profile.py
python - <<'PY'
import sys, time
sys.path.insert(0,'.')
import numpy as np
from theiasfm_amd import abi, lib, synth
from oracle import oracle
P = synth.config("venice1778_heavy")
o = abi.default_options(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR)
s = lib.Solver(P, o)
trk = abi.default_options(point_dof=3, max_num_iterations=50)
r = s.adjust_tracks(trk); s.reset()
term_d, it_d, c0_d, c1_d, ts = s.adjust_tracks(trk)
print("adjust_tracks kernel ms", ts.kernel_seconds*1e3, "iters", ts.total_iterations)
s.close()
PY
