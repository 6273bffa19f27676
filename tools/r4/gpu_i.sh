#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q > $O/r4i_tests.log 2>&1; grep -n "passed\|failed\|Error\|assert" $O/r4i_tests.log | tail -5
AIDE_ONLY=32,32,0 python tools/bench_bf16.py c5 10 2>&1 | grep "@"
AIDE_ONLY=32,64,1 python tools/bench_bf16.py c5 10 2>&1 | grep "@"
for i in 1 2 3; do
python bench.py --workload c5 --steps 30 --no-cpu-baseline --traffic none 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('c5', j['value'], j['ms_per_step'])"
done
