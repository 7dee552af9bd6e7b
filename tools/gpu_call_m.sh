#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
TMI_PROBE_PROFILE=0 python tools/scale_probe.py 1 2 4 8 > $O/r02_n_scale_probe.jsonl 2>/dev/null
TMI_PROBE_PROFILE=1 python tools/scale_probe.py 8 > $O/r02_n_scale_probe_w8_classes.jsonl 2>/dev/null
cut -c1-140 $O/r02_n_scale_probe.jsonl
