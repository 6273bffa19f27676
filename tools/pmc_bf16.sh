# usage (GPU box): bash tools/pmc_bf16.sh "<ci,co,level> ..."   -> PMC tables for the bf16 conv kernels on those layers
LAYERS=${1:-"512,512,3 128,64,0"}
for L in $LAYERS; do
  tag=$(echo $L | tr , _)
  AIDE_ONLY=$L bash tools/pmc.sh a_$tag SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC -- python $GRAFT_REPO_ROOT/tools/bench_bf16.py c5 3 >/dev/null
  AIDE_ONLY=$L bash tools/pmc.sh b_$tag SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- python $GRAFT_REPO_ROOT/tools/bench_bf16.py c5 3 >/dev/null
done
cd $GRAFT_REPO_ROOT; for f in gpurun_out/pmc_[ab]_*/*counter_collection.csv; do echo == $f; python tools/pmc_table.py $f | grep -A9 "bf16_kernel"; done
