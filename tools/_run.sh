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
nproc
for t in 16 32 64; do
echo "threads $t"
TMI_BA_HOST_THREADS=$t TMI_BA_SETUP_TIMING=1 ./tools/e2e_bench /tmp/venice_heavy.bin 10 0 3 2>&1 | grep -i "AddViews: total\|AddTracks\|observation order\|wall_seconds" | tail -4 | cut -c1-200
done
