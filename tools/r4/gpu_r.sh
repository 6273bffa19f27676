#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for v in "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=1"; do
 ok=0; bad=0
 for i in $(seq 1 14); do
  env $v AIDE_DIST_BACKEND=gloo python bench.py --gpus 8 --workload tiny --steps 3 --warmup 1 --event-steps 1 --no-cpu-baseline --traffic none > $O/r4r_tmp.out 2> $O/r4r_tmp.err
  if [ $? -eq 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); grep -h "aborting\|Error" $O/r4r_tmp.err | head -2 | cut -c1-200; fi
 done
 echo "variant [$v] ok $ok bad $bad"
done
