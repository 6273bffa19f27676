#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r4h_gputests.log 2>&1; grep -n "passed\|failed" $O/r4h_gputests.log | tail -3
python bench.py --no-cpu-baseline --traffic none 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('c2', j['value'], j['ms_per_step'])"
