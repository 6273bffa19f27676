#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_losses.py tests/test_gpu_multiclass.py -m gpu -x -q > $O/r4o_tests.log 2>&1; grep -n "passed\|failed\|Error\|assert" $O/r4o_tests.log | tail -6
