#!/bin/bash
# evidence run of a round (on the GPU box, via gpurun): tests, smoke, default bench line, kernel trace, PMC passes,
# alamo-variant kernel trace, config-5 kernel trace, scale probe.  usage: tools/evidence_run.sh <tag>   (e.g. r03_z)
T=${1:-r05_f}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
export TMI_GIT_HEAD=${TMI_GIT_HEAD:-unknown}
timeout 2400 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -8 > $O/${T}_pytest.log
grep -E "passed|failed" $O/${T}_pytest.log
python -c "import __graft_entry__ as e; e.smoke(); print('smoke ok')" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp && cd $R
rm -rf $O/${T}_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/${T}_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/${T}_pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${T}_pmc_$c -- python bench.py --steps 20 --warmup 1 --no-cpu-baseline --no-extras > $O/${T}_pmc_$c.log 2>&1
done
python tools/summarize_profile.py $T $O/${T}_stats $O/${T}_pmc_FETCH_SIZE $O/${T}_pmc_WRITE_SIZE > $O/${T}_summary.txt 2>&1
head -14 $O/${T}_summary.txt
# the reference-default operating point: kernel trace
rm -rf $O/${T}_refdef_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_refdef_stats -- python tools/refdef_probe.py 10 refdef_auto > $O/${T}_refdef.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/${T}_refdef_pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${T}_refdef_pmc_$c -- python tools/refdef_probe.py 10 refdef_auto > $O/${T}_refdef_pmc_$c.log 2>&1
done
python tools/summarize_profile.py ${T}_refdef $O/${T}_refdef_stats $O/${T}_refdef_pmc_FETCH_SIZE $O/${T}_refdef_pmc_WRITE_SIZE --workload venice1778_heavy_reference_defaults > $O/${T}_refdef_summary.txt 2>&1
head -12 $O/${T}_refdef_summary.txt
tail -1 $O/${T}_refdef.log | cut -c1-300
# the default bench line AFTER the PMC passes: pmc_latest.json now carries this build's stamp, so roofline.traffic is
# filled (the committed r04 line was taken before the passes and said traffic: null / stale)
SECONDS=0
timeout 900 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
echo "default bench.py wall: $SECONDS s"
head -c 300 $O/${T}_bench.json; echo
# the exact-solver variant (config 3): kernel trace of the alamo-sized workload
rm -rf $O/${T}_alamo_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_alamo_stats -- python bench.py --workload alamo --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/${T}_alamo_stats.log 2>&1
python tools/summarize_profile.py ${T}_alamo $O/${T}_alamo_stats --workload alamo > $O/${T}_alamo_summary.txt 2>&1
head -10 $O/${T}_alamo_summary.txt
# config 5: kernel trace
rm -rf $O/${T}_config5_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_config5_stats -- python tools/config5_classes.py 2 200 32 0 3 > $O/${T}_config5.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/${T}_config5_pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${T}_config5_pmc_$c -- python tools/config5_classes.py 2 200 32 0 3 > $O/${T}_config5_pmc_$c.log 2>&1
done
python tools/summarize_profile.py ${T}_config5 $O/${T}_config5_stats $O/${T}_config5_pmc_FETCH_SIZE $O/${T}_config5_pmc_WRITE_SIZE --workload config5 > $O/${T}_config5_summary.txt 2>&1
head -12 $O/${T}_config5_summary.txt
tail -1 $O/${T}_config5.log | cut -c1-400
mkdir -p $O/profiles_out && cp profiles/${T}_* profiles/pmc_latest.json profiles/pmc_venice1778_heavy_reference_defaults.json $O/profiles_out/ 2>/dev/null
find $O/${T}_stats $O/${T}_pmc_FETCH_SIZE $O/${T}_pmc_WRITE_SIZE $O/${T}_alamo_stats $O/${T}_config5_stats $O/${T}_refdef_stats $O/${T}_refdef_pmc_FETCH_SIZE $O/${T}_refdef_pmc_WRITE_SIZE $O/${T}_config5_pmc_FETCH_SIZE $O/${T}_config5_pmc_WRITE_SIZE -type f -size +4M -delete
TMI_PROBE_PROFILE=0 timeout 600 python tools/scale_probe.py 1 2 4 8 > $O/${T}_scale_probe.jsonl 2> $O/${T}_scale_probe.err
cut -c1-200 $O/${T}_scale_probe.jsonl
