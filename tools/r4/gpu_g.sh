#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py::test_conv3x3_winograd4_input_batchnorm tests/test_gpu_kernels.py::test_conv3x3_winograd4_fwd_dgrad tests/test_gpu_steps.py tests/test_gpu_fullsize.py::test_config3_proposed_step_256 -m gpu -x -q > $O/r4g_tests.log 2>&1; grep -n "passed\|failed\|Error\|assert" $O/r4g_tests.log | tail -8
for i in 1 2 3; do
  python bench.py --workload c3 --steps 20 --no-cpu-baseline --traffic none 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('c3', j['value'], j['ms_per_step'], j['final_loss'])"
done
