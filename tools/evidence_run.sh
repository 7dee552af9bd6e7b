#!/bin/bash
# evidence run of a round (on the GPU box, via gpurun): tests, smoke, e2e, default bench line, kernel trace,
# PMC passes, scale probe.  The tag (r02_z) names the files copied into profiles/ afterwards.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -20 > $O/r02_z_pytest.log
tail -6 $O/r02_z_pytest.log
python - <<'PY'
import sys
sys.path.insert(0,'.')
from theiasfm_amd import synth
import bench
p=synth.config("venice1778_heavy")
bench.write_problem_file(p, "/tmp/venice_heavy.bin")
PY
TMI_BA_SETUP_TIMING=1 ./tools/e2e_bench /tmp/venice_heavy.bin 10 0 2 2> $O/r02_z_e2e.err | tail -1 > $O/r02_z_e2e.json
grep -i "AddTracks\|AddViews: total" $O/r02_z_e2e.err | tail -2; cat $O/r02_z_e2e.json
python -c "import __graft_entry__ as e; e.smoke(); print('smoke ok')" 2>&1 | tail -1
SECONDS=0
timeout 900 python bench.py > $O/r02_z_bench.json 2> $O/r02_z_bench.err
echo "default bench.py wall: $SECONDS s"
tail -2 $O/r02_z_bench.err; head -c 600 $O/r02_z_bench.json; echo
cd /tmp && export TMPDIR=/tmp && cd $R
rm -rf $O/r02_z_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_z_stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/r02_z_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/r02_z_pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/r02_z_pmc_$c -- python bench.py --steps 20 --warmup 1 --no-cpu-baseline --no-extras > $O/r02_z_pmc_$c.log 2>&1
done
python tools/summarize_profile.py r02_z $O/r02_z_stats $O/r02_z_pmc_FETCH_SIZE $O/r02_z_pmc_WRITE_SIZE > $O/r02_z_summary.txt 2>&1
mkdir -p $O/profiles_out && cp profiles/r02_z_* profiles/pmc_latest.json $O/profiles_out/ 2>/dev/null
head -30 $O/r02_z_summary.txt
# keep only the small files
find $O/r02_z_stats $O/r02_z_pmc_FETCH_SIZE $O/r02_z_pmc_WRITE_SIZE -type f -size +4M -delete
TMI_PROBE_PROFILE=0 python tools/scale_probe.py 1 2 4 8 > $O/r02_z_scale_probe.jsonl 2> $O/r02_z_scale_probe.err
tail -3 $O/r02_z_scale_probe.jsonl | cut -c1-400
