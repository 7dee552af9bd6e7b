#!/bin/bash
# round 2, GPU call B: LDS-staged schur_offdiag vs the gather kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -60 > $O/r02_b_pytest.log
python bench.py --steps 20 --no-cpu-baseline --no-extras > $O/r02_b_bench.json 2> $O/r02_b_bench.err
TMI_BA_SCHUR_GATHER=1 python bench.py --steps 20 --no-cpu-baseline --no-extras > $O/r02_b_bench_gather.json 2> $O/r02_b_bench_gather.err
python bench.py --steps 20 --workload venice1778 --no-cpu-baseline --no-extras > $O/r02_b_bench_plain.json 2> $O/r02_b_bench_plain.err
TMI_BA_SCHUR_GATHER=1 python bench.py --steps 20 --workload venice1778 --no-cpu-baseline --no-extras > $O/r02_b_bench_plain_gather.json 2> $O/r02_b_bench_plain_gather.err
tail -4 $O/r02_b_pytest.log
