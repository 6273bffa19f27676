#!/bin/bash
# usage (GPU box, via gpurun): tools/gpu_round.sh <tag> [tests|notests]
# GPU tests, the default bench line, and a rocprofv3 --kernel-trace --stats run of the SAME bench command.
tag=${1:-r2}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
if [ "$2" != "notests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $R/gpurun_out/${tag}_gputests.log 2>&1
  tail -5 $R/gpurun_out/${tag}_gputests.log
fi
python bench.py > $R/gpurun_out/${tag}_bench_c2.json 2> $R/gpurun_out/${tag}_bench_c2.err
tail -c 1500 $R/gpurun_out/${tag}_bench_c2.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof_c2 -o c2 -- python $R/bench.py --no-cpu-baseline --traffic none > $R/gpurun_out/${tag}_bench_c2_under_rocprof.json 2> $R/gpurun_out/${tag}_prof_c2.log
ls $R/gpurun_out/${tag}_prof_c2 | head
