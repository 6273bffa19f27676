#!/bin/bash
# usage (GPU box): tools/r4/gpu_trace.sh <workload> [bench flags] -> gpurun_out/tr_<workload>/: kernel trace of a short run + timeline of a plain step
w=${1:-c2}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/tr_$w
rocprofv3 --kernel-trace --output-format csv -d $O/tr_$w -o $w -- python $R/bench.py --workload $w --steps 12 --warmup 3 --no-cpu-baseline --traffic none --no-kernel-events "$@" > $O/tr_$w.json 2> $O/tr_$w.log
cd $R
python tools/timeline.py $(find $O/tr_$w -name '*kernel_trace.csv') 4 --full > $O/tr_${w}_timeline.txt 2>&1
tail -5 $O/tr_${w}_timeline.txt
