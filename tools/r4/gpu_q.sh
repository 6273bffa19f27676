#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/r4q_trace_c5 -o c5 -- python $R/bench.py --workload c5 --steps 12 --warmup 3 --no-cpu-baseline --traffic none --no-kernel-events > $O/r4q_c5.json 2> $O/r4q_c5.log
cd $R
python tools/timeline.py $(find $O/r4q_trace_c5 -name '*kernel_trace.csv') 3 --full > $O/r4q_timeline_c5.txt
tail -8 $O/r4q_timeline_c5.txt
