for L in 128,64,0 256,128,1; do
  tag=$(echo $L | tr , _)
  AIDE_ONLY=$L bash tools/pmc.sh a_$tag SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT -- python $GRAFT_REPO_ROOT/tools/bench_bf16.py c5 3 >/dev/null
  AIDE_ONLY=$L bash tools/pmc.sh b_$tag SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD -- python $GRAFT_REPO_ROOT/tools/bench_bf16.py c5 3 >/dev/null
done
cd $GRAFT_REPO_ROOT; for f in gpurun_out/pmc_*/*/*counter_collection.csv gpurun_out/pmc_*/*counter_collection.csv; do [ -f $f ] && echo == $f && python tools/pmc_table.py $f | grep -A9 "bf16_kernel"; done
