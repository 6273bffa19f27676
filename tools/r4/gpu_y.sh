#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python - <<'PY'
import json, subprocess, sys, os
def run(wl, flag):
    code = "import sys, bench\nfrom aide_amd import engine\nengine.STEM_FWD[0] = %s\nsys.argv=['bench.py','--workload','%s','--no-cpu-baseline','--traffic','none'%s]\nbench.main()" % (flag, wl, ",'--steps','20'" if wl == 'c3' else '')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=os.environ['GRAFT_REPO_ROOT'])
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    return j['value']
for wl in ('c2', 'c3'):
    for i in range(4):
        print(wl, 'stem kernel %.2f   general %.2f' % (run(wl, True), run(wl, False)), flush=True)
PY
