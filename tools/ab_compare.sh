#!/bin/bash
# Same-box A/B of two engine builds (run through gpurun): theiasfm_amd/lib/ab_base.so vs ab_new.so are copied over
# the engine library in turn, three alternating bench runs each.  Build the baseline from a stash or an older commit
# with __graft_entry__.build_engine(force=True) and copy it to ab_base.so, the candidate to ab_new.so.
# usage (on the GPU box): bash tools/ab_compare.sh [workload]
cd $GRAFT_REPO_ROOT
L=theiasfm_amd/lib
W=${1:-venice1778_heavy}
for rep in 1 2 3; do
for which in base new; do
cp $L/ab_$which.so $L/libtheia_mi355_ba.so
python bench.py --workload $W --steps 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$which', round(d['ms_per_step'],3), d['pcg_iterations'], d['final_cost'], {k['kernel']:k['avg_us'] for k in d['kernels'] if k['kernel'] in ('spmv','back_substitute','point_eliminate','linearize','update_cost','camera_diag')})"
done
done
