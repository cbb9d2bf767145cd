#!/bin/bash
# A round's closing run on one box (TAG=r06): every GPU test, the smoke entry, the profile legs (kernel statistics + HBM
# counters; SQ counters for the headline's leg and the join engine's kernel), the fuzzers, then the bench line with the PMC
# files of THIS tree.  To be repeated whenever the hot path changes afterwards.
#   gpurun --timeout 5400 -- 'TAG=r06 bash tools/round_final.sh'
TAG=${TAG:-r06}
export TAG
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gpu_tests.log 2>&1
echo "gpu tests rc=$? $(grep -E 'passed|failed' gpurun_out/${TAG}_gpu_tests.log | tail -1)"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 2400 bash tools/profile_round.sh ${LEGS:-c3_cold c3 c5_cold c5 clades one_clade random identical one_species_cold one_species} 2>&1 | tail -40
cp gpurun_out/compare_*_pmc.json profiles/ 2>/dev/null
LEG=one_species timeout 600 bash tools/pmc_join.sh > gpurun_out/${TAG}_one_species_sq.txt 2>&1; tail -4 gpurun_out/${TAG}_one_species_sq.txt
LEG=one_clade KERNEL=dn_pairs TA=1 timeout 600 bash tools/pmc_join.sh > gpurun_out/${TAG}_one_clade_sq.txt 2>&1; tail -4 gpurun_out/${TAG}_one_clade_sq.txt
rm -rf gpurun_out/${TAG}_sq_*
# the fill beside the index build: a small grid of paces, and one step's timeline with the fill behind / beside the build
WGS="32 64 128 256" NAPS="0 1 4 16" STEPS=6 timeout 900 bash tools/aside_sweep.sh > gpurun_out/${TAG}_fill_aside_sweep.txt 2>&1; tail -3 gpurun_out/${TAG}_fill_aside_sweep.txt
( cd /tmp && export TMPDIR=/tmp
  for a in 0 default; do
    d=/tmp/tl_$a; rm -rf $d
    if [ $a = 0 ]; then export MASHGPU_FILL_ASIDE=0; else unset MASHGPU_FILL_ASIDE; fi
    timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $d -o p -- python $OLDPWD/tools/prof_leg.py --leg c3 --cold --steps 3 > /tmp/tl.log 2>&1
    n=beside; [ $a = 0 ] && n=behind
    python $OLDPWD/tools/step_timeline.py $d > $OLDPWD/gpurun_out/${TAG}_timeline_c3_fill_$n.txt 2>&1; tail -1 $OLDPWD/gpurun_out/${TAG}_timeline_c3_fill_$n.txt
  done )
timeout 400 python tools/compare_fuzz.py --seed 606 --seconds 150 --n 100000 > gpurun_out/${TAG}_compare_fuzz.txt 2>&1; tail -1 gpurun_out/${TAG}_compare_fuzz.txt
for w in c3 one_species; do timeout 300 python tools/ranks_on_one_gpu.py $w 8 > gpurun_out/${TAG}_ranks_$w.txt 2>/dev/null; grep '"cut"' gpurun_out/${TAG}_ranks_$w.txt; done
timeout 600 python tools/species_engines.py > gpurun_out/${TAG}_species_engines.txt 2>/dev/null; tail -2 gpurun_out/${TAG}_species_engines.txt | cut -c1-300
timeout 600 python tools/compare_e2e.py > gpurun_out/${TAG}_compare_e2e.json 2>/dev/null; tail -1 gpurun_out/${TAG}_compare_e2e.json | cut -c1-300
for w in c3 c5; do timeout 300 python tools/range_check.py $w 50000 100000 > gpurun_out/${TAG}_range_$w.txt 2>/dev/null; tail -1 gpurun_out/${TAG}_range_$w.txt; done
timeout 1500 python bench.py --detail gpurun_out/${TAG}_bench_detail.json > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_stderr.log
echo "bench rc=$?"; tail -1 gpurun_out/${TAG}_bench_line.json | cut -c1-3500
# what travels back: summaries only (64 MiB at most)
find gpurun_out -name "*kernel_trace.csv" -delete 2>/dev/null
for d in gpurun_out/${TAG}_*_fetch gpurun_out/${TAG}_*_write gpurun_out/${TAG}_*_sqa gpurun_out/${TAG}_*_sqb gpurun_out/${TAG}_*_sqc; do [ -d "$d" ] && rm -rf "$d"; done
du -sh gpurun_out | cut -f1
