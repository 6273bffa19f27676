#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1700 python -m pytest tests/test_gpu_bench.py tests/test_gpu_models.py::test_engine_assigned_gradients_keep_torch_contracts tests/test_gpu_steps.py::test_gradient_allreduce_survives_plan_eviction -m gpu -x -q -s 2>&1 | tail -25 > $O/r4d_tests.log; cat $O/r4d_tests.log
python tools/cpu_thread_sweep.py c2 > $O/r4d_cpu_threads_c2.txt 2>&1; cat $O/r4d_cpu_threads_c2.txt
