#!/bin/bash
# usage (GPU box): bash tools/pmc_mfma.sh <tag> [workload]
# MFMA-utilisation counters of the conv kernels inside the real training step (bench.py, 1 warm-up + 2 steps), two
# rocprofv3 --pmc passes (with --kernel-trace only):
#   pass a: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE
#   pass b: SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
tag=${1:-r2}; wl=${2:-c2}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_mfma_${wl}_a -o a -- python $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --traffic none > $R/gpurun_out/${tag}_pmc_mfma_${wl}_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_mfma_${wl}_b -o b -- python $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --traffic none > $R/gpurun_out/${tag}_pmc_mfma_${wl}_b.log 2>&1
cd $R
python tools/pmc_mfma_report.py $(find gpurun_out/${tag}_pmc_mfma_${wl}_a -name '*counter_collection.csv') $(find gpurun_out/${tag}_pmc_mfma_${wl}_b -name '*counter_collection.csv') > gpurun_out/${tag}_pmc_mfma_${wl}.md
cat gpurun_out/${tag}_pmc_mfma_${wl}.md
