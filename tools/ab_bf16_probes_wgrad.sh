for lib in "" abtest/lib_WG_NO_MFMA.so abtest/lib_WG_NO_FETCH.so abtest/lib_WG_NO_SHIFT.so abtest/lib_WG_NO_PUT.so; do
  echo "== ${lib:-baseline}"
  for L in 64,64,0 128,64,0 128,128,1 512,512,3 1024,512,3; do
    if [ -n "$lib" ]; then AIDE_HIP_LIB=$lib AIDE_ONLY=$L python tools/bench_bf16.py c5 5 2>&1 | grep "@" | cut -c1-22,49-80; else AIDE_ONLY=$L python tools/bench_bf16.py c5 5 2>&1 | grep "@" | cut -c1-22,49-80; fi
  done
done
