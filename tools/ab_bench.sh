#!/bin/bash
# usage (GPU box): tools/ab_bench.sh <tag> <rounds> "<env A>|<flags A>" "<env B>|<flags B>" ...
# Interleaved same-box A/B of bench.py variants (images/s per run -> gpurun_out/<tag>_ab.txt).
tag=$1; rounds=$2; shift 2
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${tag}_ab.txt; : > $out
for r in $(seq 1 $rounds); do
  i=0
  for v in "$@"; do
    envs="${v%%|*}"; flags="${v#*|}"
    line=$(env $envs python bench.py --no-cpu-baseline --traffic none $flags 2>/dev/null | tail -1)
    val=$(echo "$line" | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])" 2>/dev/null)
    echo "round $r variant $i [$v] $val" | tee -a $out
    i=$((i+1))
  done
done
