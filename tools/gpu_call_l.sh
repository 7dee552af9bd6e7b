#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -m gpu -q --tb=short -x 2>&1 | tail -3
for m in 0 1; do
if [ $m = 1 ]; then export TMI_BA_SCHUR_DEPTH1=1; fi
timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-extras > $O/r02_l_exp$m.json 2> $O/r02_l_exp$m.err
python - $m <<'PY'
import json,sys
d=json.load(open('/root/repo/gpurun_out/r02_l_exp%s.json'%sys.argv[1]))
for k in d.get('kernels',[]):
    if k['kernel'] in ('schur_offdiag',): print(sys.argv[1], k['avg_us'], k['launches'], d['ms_per_step'], d['final_cost'])
PY
done
