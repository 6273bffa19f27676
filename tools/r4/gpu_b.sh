#!/bin/bash
# wgrad bf16 ablation probes on a few C5 layer shapes (isolated kernels)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
out=$O/r4b_wg_probes.txt; : > $out
for lib in "" wg_nomfma wg_nofetch wg_noput wg_nostage wg_nobarrier wg_mfmaonly; do
  echo "== ${lib:-baseline}" >> $out
  for L in 32,32,0 64,64,0 128,64,0 128,128,1 256,128,1 512,512,3 1024,512,3; do
    if [ -n "$lib" ]; then export AIDE_HIP_LIB=$R/abtest/lib_$lib.so; else unset AIDE_HIP_LIB; fi
    AIDE_ONLY=$L python tools/bench_bf16.py c5 10 2>&1 | grep "@" | cut -c1-22,49-80 >> $out
  done
done
cat $out
