#!/bin/bash
cd $GRAFT_REPO_ROOT
for be in 15 32 64 128; do
echo -n "break_even $be: "; TMI_BA_BREAK_EVEN=$be python tools/refdef_probe.py 10 refdef_auto pb_dof4_auto 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['case'], d['ms_per_iter'], d['pcg'], d['matrix_free_its'], end=' | ')
print()"
done
