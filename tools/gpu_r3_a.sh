#!/bin/bash
# round-3 first GPU pass: GPU tests, host-time profile (direct gradients on / off), bench lines
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r3a_gputests.log 2>&1; tail -5 $O/r3a_gputests.log
python tools/host_profile.py c2 8 --profile > $O/r3a_host_c2.txt 2>&1; head -3 $O/r3a_host_c2.txt
AIDE_DIRECT_GRADS=0 python tools/host_profile.py c2 8 > $O/r3a_host_c2_plain.txt 2>&1; head -3 $O/r3a_host_c2_plain.txt
python tools/host_profile.py tiny 20 > $O/r3a_host_tiny.txt 2>&1; head -3 $O/r3a_host_tiny.txt
python bench.py > $O/r3a_bench_c2.json 2> $O/r3a_bench_c2.err; tail -c 400 $O/r3a_bench_c2.json
AIDE_DIRECT_GRADS=0 python bench.py --no-cpu-baseline --traffic none > $O/r3a_bench_c2_plain.json 2>/dev/null; head -c 300 $O/r3a_bench_c2_plain.json
python bench.py --no-cpu-baseline --traffic none > $O/r3a_bench_c2_b.json 2>/dev/null; head -c 300 $O/r3a_bench_c2_b.json
python bench.py --workload c2 --precision bf16 --no-cpu-baseline --traffic none > $O/r3a_bench_c2_bf16.json 2>/dev/null; head -c 300 $O/r3a_bench_c2_bf16.json
python bench.py --workload c3 --steps 20 > $O/r3a_bench_c3.json 2> $O/r3a_bench_c3.err; tail -c 600 $O/r3a_bench_c3.json
python bench.py --workload c5 --no-cpu-baseline --traffic none > $O/r3a_bench_c5.json 2>/dev/null; head -c 300 $O/r3a_bench_c5.json
python tools/phase_trace.py c2 > $O/r3a_phase_c2.txt 2>&1
