#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
for m in 0 4; do
TMI_BA_EXP=$m timeout 300 python bench.py --steps 6 --no-cpu-baseline --no-extras > $O/r02_l_exp$m.json 2> $O/r02_l_exp$m.err
python - $m <<'PY'
import json,sys
d=json.load(open('/root/repo/gpurun_out/r02_l_exp%s.json'%sys.argv[1]))
for k in d.get('kernels',[]):
    if k['kernel'] in ('schur_offdiag',): print(sys.argv[1], k['avg_us'], k['launches'], d['ms_per_step'], d['final_cost'])
PY
done
