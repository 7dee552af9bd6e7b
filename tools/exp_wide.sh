#!/bin/bash
# GPU experiment: threshold of the 16-lanes-per-track slices (TMI_BA_WIDE_K), worlds 1 and 8
for k in ${@:-12 16 24 32}; do
  echo "== TMI_BA_WIDE_K=$k"
  TMI_BA_WIDE_K=$k python tools/scale_probe.py 1 8 2>/dev/null | grep world | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l)
    if d['schur_mode']=='explicit' and d['world']>1: continue
    print(d['world'], d['schur_mode'], d['ms_per_iter'], d['pcg'], {k:round(v[1]/v[0]*1e3) for k,v in d['kernels'].items()})"
done
