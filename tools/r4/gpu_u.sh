#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "stem" > $O/r4u_k.log 2>&1; tail -5 $O/r4u_k.log
timeout 300 python tools/bench_stem.py 2>&1 | tee $O/r4u_stem.txt
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_steps.py tests/test_gpu_bf16.py -m gpu -x -q > $O/r4u_tests.log 2>&1; grep -n "passed\|failed\|Error\|assert" $O/r4u_tests.log | tail -6
python - <<'PY'
import json, subprocess, sys, os
def run(wl, flag):
    code = "import sys, bench\nfrom aide_amd import engine\nengine.STEM_FWD[0] = %s\nsys.argv=['bench.py','--workload','%s','--no-cpu-baseline','--traffic','none']\nbench.main()" % (flag, wl)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=os.environ['GRAFT_REPO_ROOT'])
    try:
        j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    except Exception:
        print(r.stderr[-2000:]); raise
    return j['value'], j['ms_per_step']
for wl in ('c2', 'c5', 'c4', 'c3'):
    for i in range(2):
        print(wl, 'stem kernel', run(wl, 'True'), ' general', run(wl, 'False'), flush=True)
PY
