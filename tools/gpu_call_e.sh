#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_host_shim.py tests/test_gpu_tracks.py tests/test_two_view.py -m gpu -q --tb=short 2>&1 | tail -30 > $O/r02_e_pytest.log
tail -8 $O/r02_e_pytest.log
python - <<'PY'
import sys, time
sys.path.insert(0,'.')
from theiasfm_amd import synth
import bench
p=synth.config("venice1778_heavy")
bench.write_problem_file(p, "/tmp/venice_heavy.bin")
PY
TMI_BA_SETUP_TIMING=1 ./tools/e2e_bench /tmp/venice_heavy.bin 10 0 1 2>&1 | tail -30
python - <<'PY'
import time, numpy as np
from theiasfm_amd import abi, lib, synth
prob = synth.config("venice1778_heavy")
o = abi.default_options(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, max_num_iterations=2, use_inner_iterations=0)
s = lib.Solver(prob, o)
s.solve(o)
for rep in range(3):
    f, m, fs = s.filter_outlier_tracks(4.0, 2.0)
    print("filter call_ms", fs.seconds*1e3, "kernel_us", fs.kernel_seconds*1e6)
for rep in range(3):
    sel, ln, err, ss = s.select_good_tracks(10, 100, 100)
    print("select call_ms", ss.seconds*1e3)
PY
