#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -40 > $O/r02_f_pytest.log
tail -12 $O/r02_f_pytest.log
python bench.py --steps 20 --no-cpu-baseline --no-extras > $O/r02_f_bench.json 2> $O/r02_f_bench.err
TMI_BA_PCG_LEGACY=1 python bench.py --steps 20 --no-cpu-baseline --no-extras > $O/r02_f_bench_legacy_pcg.json 2> $O/r02_f_bench_legacy_pcg.err
python tools/scale_probe.py 1 8 > $O/r02_f_scale_probe.jsonl 2> $O/r02_f_scale_probe.err
TMI_BA_PCG_LEGACY=1 python tools/scale_probe.py 8 > $O/r02_f_scale_probe_legacy.jsonl 2>&1
