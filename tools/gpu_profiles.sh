#!/bin/bash
# usage (GPU box): tools/gpu_profiles.sh <tag>   -> gpurun_out/<tag>_*: bench lines (c2 default, c5, c4, c3), rocprofv3
# --kernel-trace --stats of the SAME default bench command, MFMA PMC tables
tag=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python bench.py > $O/${tag}_bench_c2.json 2> $O/${tag}_bench_c2.err; tail -c 600 $O/${tag}_bench_c2.json
python bench.py --workload c5 > $O/${tag}_bench_c5.json 2> $O/${tag}_bench_c5.err; tail -c 300 $O/${tag}_bench_c5.json
python bench.py --workload c4 --no-cpu-baseline --traffic none > $O/${tag}_bench_c4.json 2>/dev/null
python bench.py --workload c3 --steps 20 > $O/${tag}_bench_c3.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_c2 -o c2 -- python $R/bench.py --no-cpu-baseline --traffic none > $O/${tag}_bench_c2_under_rocprof.json 2> $O/${tag}_prof_c2.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_c5 -o c5 -- python $R/bench.py --workload c5 --no-cpu-baseline --traffic none > $O/${tag}_bench_c5_under_rocprof.json 2> $O/${tag}_prof_c5.log
cd $R
bash tools/pmc_mfma.sh $tag c2 > /dev/null 2>&1
bash tools/pmc_mfma.sh $tag c5 > /dev/null 2>&1
ls $O | grep $tag
