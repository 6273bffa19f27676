#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1700 python -m pytest tests -m gpu -x -q > $O/r4s_gputests.log 2>&1; grep -n "passed\|failed" $O/r4s_gputests.log | tail -2
for i in 1 2 3; do python bench.py --workload c5 --steps 30 --no-cpu-baseline --traffic none 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('c5', j['value'], j['ms_per_step'])"; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/r4s_trace_c5 -o c5 -- python $R/bench.py --workload c5 --steps 12 --warmup 3 --no-cpu-baseline --traffic none --no-kernel-events > $O/r4s_c5.json 2> $O/r4s_c5.log
cd $R
python tools/timeline.py $(find $O/r4s_trace_c5 -name '*kernel_trace.csv') 3 --full > $O/r4s_timeline_c5.txt
head -8 $O/r4s_timeline_c5.txt | cut -c1-100
