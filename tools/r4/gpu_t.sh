#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_steps.py tests/test_gpu_fullsize.py::test_config3_proposed_step_256 -m gpu -x -q > $O/r4t_tests.log 2>&1; grep -n "passed\|failed\|Error\|assert" $O/r4t_tests.log | tail -6
python - <<'PY'
import json, subprocess, sys, os
def run(flag):
    code = "import sys, bench\nfrom aide_amd import engine\nengine.GROUPED_BN[0] = %s\nsys.argv=['bench.py','--workload','c3','--steps','20','--no-cpu-baseline','--traffic','none']\nbench.main()" % flag
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=os.environ['GRAFT_REPO_ROOT'])
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    return j['value'], j['ms_per_step']
for i in range(2):
    print('c3 grouped', run('True'), ' per-group', run('False'))
PY
