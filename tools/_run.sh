#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys
sys.path.insert(0,'.')
from theiasfm_amd import synth
import bench
p=synth.config("venice1778_heavy")
bench.write_problem_file(p, "/tmp/venice_heavy.bin")
PY
./tools/e2e_bench /tmp/venice_heavy.bin 10 0 3 2>/dev/null | tail -1 | cut -c150-420
./tools/e2e_bench /tmp/venice_heavy.bin 10 1 3 2>/dev/null | tail -1 | cut -c150-420
