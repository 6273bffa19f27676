#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
bash tools/ab_bench.sh r4p_c2 2 "|" "AIDE_HIP_LIB=$R/abtest/lib_g4_t112.so|--allow-probes" "AIDE_HIP_LIB=$R/abtest/lib_g4_t144.so|--allow-probes" "AIDE_HIP_LIB=$R/abtest/lib_g4_t160.so|--allow-probes"
bash tools/ab_bench.sh r4p_c4 2 "|--workload c4" "AIDE_HIP_LIB=$R/abtest/lib_g4_t144.so|--workload c4 --allow-probes" "AIDE_HIP_LIB=$R/abtest/lib_g4_t160.so|--workload c4 --allow-probes"
