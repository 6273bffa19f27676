#!/bin/bash
# refresh the c3 artefacts of a profile set: tools/r4/gpu_c3prof.sh <tag>
tag=${1:-r04}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python bench.py --workload c3 --steps 20 > $O/${tag}_bench_c3.json 2> $O/${tag}_bench_c3.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${tag}_prof_c3
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_c3 -o c3 -- python $R/bench.py --workload c3 --steps 20 --no-cpu-baseline --traffic none > $O/${tag}_bench_c3_under_rocprof.json 2> $O/${tag}_prof_c3.log
head -c 300 $O/${tag}_bench_c3.json; echo; head -c 300 $O/${tag}_bench_c3_under_rocprof.json
