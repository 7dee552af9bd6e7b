#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "select or filter or track or shim" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -5
cd /tmp && export TMPDIR=/tmp && cd $R
rm -rf $O/r02_m_sel
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_m_sel -- python tools/select_profile.py > $O/r02_m_sel.log 2>&1
grep "call ms" $O/r02_m_sel.log
python - <<'PY'
import csv, glob
f = glob.glob('/root/repo/gpurun_out/r02_m_sel/**/*_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    n = r["Name"]
    if ("select" in n or "track_stats" in n or "scatter" in n) and "rocprim" not in n:
        print(n.split("(")[0][:60], r["Calls"], "%.1f us avg" % (float(r["AverageNs"])/1e3))
PY
find $O/r02_m_sel -type f -size +2M -delete
python tools/select_profile.py | tail -2
