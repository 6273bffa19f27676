#!/bin/bash
# usage (GPU box): bash tools/fetch_calib.sh  -> gpurun_out/fetch_calib.txt: FETCH_SIZE / WRITE_SIZE (raw KB) per pattern vs 2 GiB
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/ubench/fetch_calib.hip -o /tmp/fetch_calib || exit 1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/fc_$c; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/fc_$c -o fc -- /tmp/fetch_calib > /tmp/fc_$c.log 2>&1
done
python3 - <<'PY' | tee $O/fetch_calib.txt
import csv, glob, re, collections
B = float(2 << 30)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    agg = collections.OrderedDict()
    for f in glob.glob('/tmp/fc_%s/**/*counter_collection.csv' % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != c: continue
            m = re.search(r'seg_kernel<(\d+), (true|false|0|1)>', r['Kernel_Name'])
            if not m: continue
            a = agg.setdefault((int(m.group(1)), m.group(2) in ('true', '1')), [])
            a.append(float(r['Counter_Value']) * 1e3)
    for (seg, wr), v in sorted(agg.items()):
        print('%-10s seg %4d B  %-5s  counter / bytes moved = %.3f  (n=%d, raw %.1f MB)' % (c, seg, 'write' if wr else 'read', sum(v) / len(v) / B, len(v), sum(v) / len(v) / 1e6))
PY
