#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -5
TMI_PROBE_PROFILE=0 python tools/scale_probe.py 1 2 4 8 > $O/r02_m_scale_probe.jsonl 2> $O/r02_m_scale_probe.err
cut -c1-200 $O/r02_m_scale_probe.jsonl
