#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_setup.py -m gpu -q --tb=short -x 2>&1 | tail -40 > $O/r02_i_setup.log
tail -30 $O/r02_i_setup.log
python - <<'PY'
import sys
sys.path.insert(0,'.')
from theiasfm_amd import synth
import bench
p=synth.config("venice1778_heavy")
bench.write_problem_file(p, "/tmp/venice_heavy.bin")
PY
TMI_BA_SETUP_TIMING=1 ./tools/e2e_bench /tmp/venice_heavy.bin 10 0 2 2>&1 | tail -24
