#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -60 > $O/r02_d_pytest.log
tail -25 $O/r02_d_pytest.log
python bench.py --steps 20 > $O/r02_d_bench.json 2> $O/r02_d_bench.err
tail -3 $O/r02_d_bench.err
