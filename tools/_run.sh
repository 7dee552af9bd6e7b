#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/auto_probe.py 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | grep auto
for w in venice1778_heavy venice1778; do
for m in auto explicit; do
python bench.py --workload $w --steps 20 --no-cpu-baseline --no-extras --schur-mode $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$w $m', round(d['ms_per_step'],3), d['pcg_iterations'], d['final_cost'], d.get('matrix_free_lm_iterations_in_last_solve'), d['roofline']['kernel'], d['roofline']['frac'])"
done; done
