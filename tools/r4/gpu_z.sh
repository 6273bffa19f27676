#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 2400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_steps.py tests/test_gpu_bf16.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/r4z_tests.log 2>&1; grep -n "passed\|failed\|Error\|assert" $O/r4z_tests.log | tail -6
python - <<'PY'
import json, subprocess, sys, os
def run(wl, flag):
    code = "import sys, bench\nfrom aide_amd import engine\nengine.HANDOVER_ON_KERNEL[0] = %s\nsys.argv=['bench.py','--workload','%s','--no-cpu-baseline','--traffic','none'%s]\nbench.main()" % (flag, wl, ",'--steps','20'" if wl == 'c3' else '')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=os.environ['GRAFT_REPO_ROOT'])
    try:
        j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    except Exception:
        print(r.stderr[-2000:]); raise
    return j['value']
for wl in ('c2', 'c4', 'c5', 'c3'):
    for i in range(3):
        print(wl, 'event on the kernel %.2f   record packet %.2f' % (run(wl, True), run(wl, False)), flush=True)
PY
