#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8
cd /tmp && export TMPDIR=/tmp && cd $R
rm -rf $O/r02_m_alamo
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_m_alamo -- python bench.py --workload alamo --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/r02_m_alamo.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/root/repo/gpurun_out/r02_m_alamo/**/*_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:8]:
    print(r["Name"].split("(")[0][:64], r["Calls"], "%.1f us avg" % (float(r["AverageNs"])/1e3), "%.2f ms" % (float(r["TotalDurationNs"])/1e6), r["Percentage"])
PY
find $O/r02_m_alamo -type f -size +2M -delete
python bench.py --workload alamo --steps 10 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('alamo ms/step', d['ms_per_step'], d['final_rmse'], [ (k['kernel'],k['avg_us']) for k in d['kernels'] if k['kernel']=='cholesky'])"
