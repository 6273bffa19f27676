#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -k "wgrad_winograd4" -m gpu -x -q 2>&1 | grep "passed\|failed"
for L in "128 64 256" "512 256 64" "64 64 256" "1024 512 32"; do python tools/ab_one.py wgrad4 $L 2>/dev/null | tail -1; done
bash tools/pmc.sh r4m SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -- python $R/tools/ab_one.py wgrad4 128 64 256 > /dev/null 2>&1
python - <<'PY'
import csv, glob, os, collections
f = glob.glob(os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out', 'pmc_r4m', '**', '*counter_collection.csv'), recursive=True)
acc = collections.defaultdict(lambda: [0.0, 0])
for p in f:
    for r in csv.DictReader(open(p)):
        if 'wgrad4_kernel' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
print('PMC ' + '  '.join('%s %.3e' % (k, v[0] / max(v[1], 1)) for k, v in sorted(acc.items())))
PY
for i in 1 2 3; do python bench.py --no-cpu-baseline --traffic none 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('c2', j['value'], j['ms_per_step'])"; done
