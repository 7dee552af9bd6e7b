#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_two_view.py tests/test_ceres_golden.py -m gpu -q --tb=short 2>&1 | tail -60 > $O/r02_c_pytest.log
tail -30 $O/r02_c_pytest.log
python - <<'PY' 2>&1 | tail -20
import time, numpy as np
from theiasfm_amd import abi, lib, synth
B = synth.make_two_view_batch(20000, 5, max_corr=300)
for rep in range(2):
    D = B.copy()
    t=time.time(); term, it, c0, c1, ts = lib.adjust_two_views(D, 4); el=time.time()-t
    print("pairs", B.num_pairs, "corr", int(B.correspondence_ptr[-1]), "kernel_ms", ts.kernel_seconds*1e3, "call_s", el, "iters", ts.total_iterations, "success", ts.num_success)
PY
