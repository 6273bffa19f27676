#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "stem" > $O/r4v_k.log 2>&1; tail -5 $O/r4v_k.log
timeout 300 python tools/bench_stem.py 2>&1 | tee $O/r4v_stem.txt
