cd $GRAFT_REPO_ROOT
for lib in "" abtest/lib_WS_NO_GSTORE.so abtest/lib_WS_NO_PUT.so abtest/lib_WS_NO_FETCH.so abtest/lib_WS_NO_MFMA.so abtest/lib_WS_NO_DUMP.so; do
  echo "== ${lib:-ws baseline}"
  for L in 64,64,0 128,64,0; do
    if [ -n "$lib" ]; then AIDE_BF16_WS=2 AIDE_HIP_LIB=$lib AIDE_ONLY=$L python tools/bench_bf16.py c5 10 2>&1 | grep "@" | cut -c1-46; else AIDE_BF16_WS=2 AIDE_ONLY=$L python tools/bench_bf16.py c5 10 2>&1 | grep "@" | cut -c1-46; fi
  done
done
