#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
bash tools/ab_bench.sh r4n_c2 3 "|" "AIDE_HIP_LIB=$R/abtest/lib_g4_old.so|--allow-probes"
bash tools/ab_bench.sh r4n_c4 2 "|--workload c4" "AIDE_HIP_LIB=$R/abtest/lib_g4_old.so|--workload c4 --allow-probes"
bash tools/ab_bench.sh r4n_c3 2 "|--workload c3 --steps 20" "AIDE_HIP_LIB=$R/abtest/lib_g4_old.so|--workload c3 --steps 20 --allow-probes"
