#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_steps.py tests/test_gpu_fullsize.py::test_config3_proposed_step_256 tests/test_gpu_kernels.py -m gpu -x -q > $O/r4e_tests.log 2>&1; grep -n "passed\|failed\|Error" $O/r4e_tests.log | tail -5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4e_prof_c3 -o c3 -- python $R/bench.py --workload c3 --steps 20 --no-cpu-baseline --traffic none > $O/r4e_c3_under_rocprof.json 2> $O/r4e_prof_c3.log
python - <<'PY'
import csv,glob,os
f=glob.glob(os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out/r4e_prof_c3/**/*kernel_stats.csv'),recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms',tot/1e6)
for r in rows[:22]:
    print('%6d %9.1f %8.1f %5.1f%%  %s'%(int(r['Calls']),float(r['TotalDurationNs'])/1e6,float(r['AverageNs'])/1e3,float(r['Percentage']),r['Name'].replace('(anonymous namespace)::','')[:90]))
PY
