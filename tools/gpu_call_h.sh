#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
for i in 1 2 3; do timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -15 > $O/r02_h_pytest_$i.log; grep -h "passed\|failed" $O/r02_h_pytest_$i.log; done
for i in 1 2; do TMI_BA_PCG_SPEC=1 timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -15 > $O/r02_h_pytest_spec_$i.log; grep -h "passed\|failed" $O/r02_h_pytest_spec_$i.log; done
grep -h "AssertionError\|FAILED" $O/r02_h_pytest_*.log | head
