#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -k upsample -m gpu -x -q 2>&1 | grep "passed\|failed"
for i in 1 2 3; do
python bench.py --workload c5 --steps 30 --no-cpu-baseline --traffic none 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('c5', j['value'], j['ms_per_step'])"
done
python bench.py --no-cpu-baseline --traffic none 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('c2', j['value'], j['ms_per_step'])"
python tools/bench_spatial.py 2>&1 | tail -12
