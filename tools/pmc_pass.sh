#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <kernel-regex> <counter> [<counter>...]   (one pass, averaged per kernel)
tag=$1; re=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$tag
rm -rf $O; cd $R
timeout 500 rocprofv3 --kernel-trace --pmc "$@" --kernel-include-regex "$re" --output-format csv -d $O -- python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras > $O.log 2>&1
python - "$O" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)
if not f:
    print("no counter file"); sys.exit(0)
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
for r in csv.DictReader(open(f[0])):
    k = (r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])
    acc[k] += float(r["Counter_Value"]); cnt[k] += 1
for k in sorted(acc):
    print(k[0], k[1], "avg/launch %.4g" % (acc[k] / cnt[k]), "launches", cnt[k])
PY
