#!/bin/bash
# round 4, first GPU pass: baselines of this round's box + A/B of the existing bf16 weight-gradient switches on the C5 step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
python bench.py --no-cpu-baseline --traffic none > $O/r4a_c2.json 2> $O/r4a_c2.err; tail -c 400 $O/r4a_c2.json | head -c 200; echo
bash tools/ab_bench.sh r4a_c5 2 "|--workload c5 --steps 30" \
  "AIDE_BF16_WG_R=2|--workload c5 --steps 30" \
  "AIDE_BF16_WG_R=2 AIDE_BF16_WG_TARGET=384|--workload c5 --steps 30" \
  "AIDE_BF16_WG_R=2 AIDE_BF16_WG_TARGET=256|--workload c5 --steps 30" \
  "AIDE_BF16_WG_NWCO_MINFLOPS=5e10|--workload c5 --steps 30" \
  "AIDE_BF16_WG_NWCO=2|--workload c5 --steps 30" \
  "AIDE_BF16_WG_TARGET=256|--workload c5 --steps 30"
AIDE_BF16_WG_R=2 AIDE_BF16_WG_TARGET=384 python tools/bench_bf16.py c5 10 > $O/r4a_layers_c5_R2.txt 2>&1
python tools/bench_bf16.py c5 10 > $O/r4a_layers_c5.txt 2>&1
tail -4 $O/r4a_layers_c5_R2.txt $O/r4a_layers_c5.txt
