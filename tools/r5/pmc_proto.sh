#!/bin/bash
# usage (GPU box): bash tools/r5/pmc_proto.sh <tag> -- issue / wait / LDS counters of the split-bf16 F(4x4) prototype and the fp32
# F(4x4) kernel on the two prototype layers (tools/proto_wino4s.py --quick), two rocprofv3 --pmc passes
tag=${1:-r5proto}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_a -o a -- python $R/tools/proto_wino4s.py --quick > $R/gpurun_out/${tag}_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_b -o b -- python $R/tools/proto_wino4s.py --quick > $R/gpurun_out/${tag}_b.log 2>&1
cd $R
python tools/pmc_mfma_report.py $(find gpurun_out/${tag}_a -name '*counter_collection.csv') $(find gpurun_out/${tag}_b -name '*counter_collection.csv') > gpurun_out/${tag}.md
cat gpurun_out/${tag}.md
