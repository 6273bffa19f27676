#!/bin/bash
# round-3 second GPU pass: bf16 epilogue statistics (tests + A/B), host time after the fixes, C5 per-layer timeline + layer sweep
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_models.py tests/test_gpu_losses.py tests/test_gpu_steps.py -x -q > $O/r3b_tests.log 2>&1; tail -3 $O/r3b_tests.log
python tools/host_profile.py c2 8 --profile > $O/r3b_host_c2.txt 2>&1; head -3 $O/r3b_host_c2.txt | cut -c1-400
python tools/host_profile.py tiny 20 > $O/r3b_host_tiny.txt 2>&1; head -3 $O/r3b_host_tiny.txt | cut -c1-400
python bench.py --workload c2 --precision bf16 --no-cpu-baseline --traffic none > $O/r3b_bench_c2_bf16.json 2>/dev/null; head -c 250 $O/r3b_bench_c2_bf16.json; echo
bash tools/ab_bench.sh r3b 2 "AIDE_BF16_EPILOGUE_STATS=0|--workload c5" "AIDE_BF16_EPILOGUE_STATS=1|--workload c5"
python tools/phase_trace.py c5 > $O/r3b_phase_c5.txt 2>&1
python tools/bench_bf16.py c5 10 > $O/r3b_sweep_c5.txt 2>&1; cat $O/r3b_sweep_c5.txt
