#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6
for e in 0 1; do
if [ $e = 1 ]; then export TMI_BA_NO_ADAPTIVE=1; fi
python bench.py --steps 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('no_adaptive=$e', d['ms_per_step'], d['pcg_iterations'], d['final_cost'], {k['kernel']:(k['launches'],k['avg_us']) for k in d['kernels'] if k['kernel'] in ('schur_offdiag','spmv','point_eliminate')})"
python bench.py --workload venice1778 --steps 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('plain no_adaptive=$e', d['ms_per_step'], d['pcg_iterations'], d['final_cost'])"
done
