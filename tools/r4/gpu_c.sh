#!/bin/bash
# GPU suite + headline benches after the prune / stateless-ABI refactor
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r4c_gputests.log 2>&1; tail -15 $O/r4c_gputests.log
for w in c2 c3 c5; do
  st=""; [ $w == c3 ] && st="--steps 20"
  python bench.py --workload $w $st --no-cpu-baseline --traffic none 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$w', j['value'], j['ms_per_step'])"
done
