#!/bin/bash
# round 2, GPU call A: parity suite, the bench line, A/B of the evaluation paths, kernel-trace stats
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -80 > $O/r02_a_pytest.log
python bench.py --steps 20 --warmup 2 > $O/r02_a_bench.json 2> $O/r02_a_bench.err
TMI_BA_LEGACY_EVAL=1 python bench.py --steps 20 --no-cpu-baseline --no-extras > $O/r02_a_bench_legacy.json 2> $O/r02_a_bench_legacy.err
TMI_BA_LINEARIZE_OCC1=1 python bench.py --steps 20 --no-cpu-baseline --no-extras > $O/r02_a_bench_occ1.json 2> $O/r02_a_bench_occ1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_a_stats -- python $R/bench.py --steps 10 --no-cpu-baseline --no-extras > $O/r02_a_stats.log 2>&1
cd $R
find $O/r02_a_stats -name "*_kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r02_a_kernel_stats.csv
rm -rf $O/r02_a_stats/*/*kernel_trace.csv 2>/dev/null
tail -5 $O/r02_a_pytest.log
