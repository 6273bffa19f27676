#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python - <<'PY'
import json, subprocess, sys, os
def run(wl, first, every, mf):
    code = "import sys, bench\nfrom aide_amd import engine\nengine.WGRAD_HANDOVER[:] = [%d, %d, %g]\nsys.argv=['bench.py','--workload','%s','--no-cpu-baseline','--traffic','none'%s]\nbench.main()" % (first, every, mf, wl, ",'--steps','20'" if wl == 'c3' else '')
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=os.environ['GRAFT_REPO_ROOT'])
    try:
        j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    except Exception:
        print(r.stderr[-2000:]); raise
    return j['value']
for wl in ("c2", "c4", "c5"):
    for i in range(2):
        print(wl, ' '.join('%s:%.2f' % (k, run(wl, *k)) for k in ((0, 1, 1e30), (6, 2, 1e30), (6, 2, 2e10), (6, 2, 4e10), (4, 2, 4e10), (8, 2, 4e10))), flush=True)
PY
