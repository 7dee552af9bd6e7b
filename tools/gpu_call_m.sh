#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_setup.py tests/test_gpu_sharded.py -m gpu -q --tb=short -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -12
TMI_BA_SETUP_TIMING=1 TMI_PROBE_PROFILE=0 python tools/scale_probe.py 8 2>&1 | grep -i "total\|world" | cut -c1-150 | head -8
