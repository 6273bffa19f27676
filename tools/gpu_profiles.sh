#!/bin/bash
# usage (GPU box): tools/gpu_profiles.sh <tag> [tests]  -> gpurun_out/<tag>_*: bench lines of every BASELINE configuration
# (c2 default, c3, c4, c5: roofline + cpu_baseline + live traffic), rocprofv3 --kernel-trace --stats of the SAME commands,
# MFMA PMC tables, whole-step HBM traffic tables (FETCH_SIZE / WRITE_SIZE in separate passes), phase trace, host profile
tag=${1:-r03}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
if [ "$2" == "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/${tag}_gputests.log 2>&1; tail -2 $O/${tag}_gputests.log
fi
python bench.py > $O/${tag}_bench_c2.json 2> $O/${tag}_bench_c2.err; tail -c 300 $O/${tag}_bench_c2.json
python bench.py --workload c3 --steps 20 > $O/${tag}_bench_c3.json 2> $O/${tag}_bench_c3.err
python bench.py --workload c4 > $O/${tag}_bench_c4.json 2> $O/${tag}_bench_c4.err
python bench.py --workload c5 > $O/${tag}_bench_c5.json 2> $O/${tag}_bench_c5.err
for w in c3 c4 c5; do head -c 200 $O/${tag}_bench_$w.json; echo; done
cd /tmp && export TMPDIR=/tmp
for w in c2 c3 c4 c5; do
  st=""; [ $w == c3 ] && st="--steps 20"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_$w -o $w -- python $R/bench.py --workload $w $st --no-cpu-baseline --traffic none > $O/${tag}_bench_${w}_under_rocprof.json 2> $O/${tag}_prof_$w.log
done
cd $R
bash tools/pmc_mfma.sh $tag c2 > /dev/null 2>&1
bash tools/pmc_mfma.sh $tag c5 > /dev/null 2>&1
for w in c2 c5; do
  bash tools/pmc.sh ${tag}_${w}_fetch FETCH_SIZE -- python $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --traffic none > /dev/null 2>&1
  bash tools/pmc.sh ${tag}_${w}_write WRITE_SIZE -- python $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --traffic none > /dev/null 2>&1
  python tools/traffic_report.py $(find $O/pmc_${tag}_${w}_fetch -name '*counter_collection.csv') $(find $O/pmc_${tag}_${w}_write -name '*counter_collection.csv') 3 \
      $O/${tag}_hbm_traffic_$w.md $O/${tag}_traffic_$w.json "$w step" > /dev/null 2>&1
done
for w in c2 c5; do bash tools/r4/gpu_trace.sh $w > /dev/null 2>&1; cp $O/tr_${w}_timeline.txt $O/${tag}_timeline_$w.txt; done
python tools/phase_trace.py c2 > $O/${tag}_phase_c2.txt 2>&1
python tools/host_profile.py c2 8 > $O/${tag}_host_c2.txt 2>&1
ls $O | grep "^${tag}_" | tr '\n' ' '
