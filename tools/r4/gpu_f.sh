#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for i in 1 2; do
for v in "" "XPROBE_SKIP_LAZY=1"; do
  env $v python - <<'PY'
import os, sys, json, subprocess
sys.argv=['bench.py','--workload','c3','--steps','20','--no-cpu-baseline','--traffic','none']
import bench
bench.KNOWN_SWITCHES = bench.KNOWN_SWITCHES + ('XPROBE_SKIP_LAZY',)
import io, contextlib
buf=io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
j=json.loads(buf.getvalue().strip().split('\n')[-1])
print('c3', os.environ.get('XPROBE_SKIP_LAZY'), j['value'], j['ms_per_step'], j['final_loss'])
PY
done; done
