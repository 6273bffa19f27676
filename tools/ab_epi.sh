cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r3c_gputests.log 2>&1; grep -E "passed|failed" $O/r3c_gputests.log | tail -2
echo "== layer sweeps (fp32 F4 / direct), base then new"
AIDE_HIP_LIB=abtest/lib_BASE.so python tools/bench_conv.py wino4 2>&1 | grep -v amdgpu | cut -c1-110 > $O/r3c_conv_base.txt
python tools/bench_conv.py wino4 2>&1 | grep -v amdgpu | cut -c1-110 > $O/r3c_conv_new.txt
paste -d'\n' $O/r3c_conv_base.txt $O/r3c_conv_new.txt | tail -44
AIDE_HIP_LIB=abtest/lib_BASE.so python tools/bench_conv.py fwd 2>&1 | grep "total" ; python tools/bench_conv.py fwd 2>&1 | grep "total"
echo "== bf16 sweep base / new(ws=1) / new(ws=0)"
AIDE_HIP_LIB=abtest/lib_BASE.so python tools/bench_bf16.py c5 10 2>&1 | grep -v amdgpu | cut -c1-58
python tools/bench_bf16.py c5 10 2>&1 | grep -v amdgpu | cut -c1-58
AIDE_BF16_WS=0 python tools/bench_bf16.py c5 10 2>&1 | grep -v amdgpu | cut -c1-58
bash tools/ab_bench.sh r3c 2 "AIDE_HIP_LIB=abtest/lib_BASE.so|" "AIDE_X=1|" "AIDE_HIP_LIB=abtest/lib_BASE.so|--workload c5" "AIDE_X=1|--workload c5" "AIDE_BF16_WS=0|--workload c5"
