"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
usage: traffic_report.py <fetch counter_collection.csv> <write counter_collection.csv> <steps run> <out.md> <out.json> [title]
Units: counters are KB; FETCH_SIZE is x2-corrected for 16 B/lane reads (MI355X_MICROARCH.md, HBM section; calibrated
on adam_kernel below)."""
import csv, sys, re, json, collections
fetch_csv, write_csv, steps, out_md, out_json = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4], sys.argv[5]


def load(path, counter):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).replace('void ', '')
        n = re.sub(r'\((?:[^()]|\([^()]*\))*\)$', '', n)[:70]
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1; a[1] += float(r['Counter_Value'])
    return agg


F, W = load(fetch_csv, 'FETCH_SIZE'), load(write_csv, 'WRITE_SIZE')
# operator families as bench.py tags them: the main kernel (+ its split reduce on split layers)
fam = {'conv3x3_wino4_kernel': ('conv3x3_wino4_kernel', 'w4_splitk_reduce_kernel'),
       'conv3x3_wino_kernel': ('conv3x3_wino_kernel', 'wino_splitk_reduce_kernel'),
       'conv3x3_mfma_kernel': ('conv3x3_mfma_kernel', '`splitk_reduce_kernel'),
       'conv3x3_wgrad4_kernel': ('conv3x3_wgrad4_kernel',),
       'conv3x3_bf16_kernel': ('conv3x3_bf16_kernel', 'bf16_splitk_reduce_kernel'),
       'conv3x3_wgrad_bf16_kernel': ('conv3x3_wgrad_bf16_kernel',)}
main = {k: v[0] for k, v in fam.items()}
title = sys.argv[6] if len(sys.argv) > 6 else 'FuseUNet C2 step'
lines = ['# HBM traffic per kernel, ' + title + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)', '',
         'Raw counter units are KiB (tools/fetch_calib.sh). FETCH_SIZE reports 1/2 of the bytes of 16 B/lane reads on gfx950, down to 64-byte segments',
         '(MI355X_MICROARCH.md, HBM section): the x2 column applies that correction; adam_kernel (20 B/param read, 16 B/param',
         'written, 26.68 M params = 533.5 / 426.8 MB) is the calibration row.', '',
         '| kernel | launches/step | FETCH raw MB/launch | x2 corrected | WRITE MB/launch |', '|---|---|---|---|---|']
tf = tw = 0.0
famtot = {k: [0.0, 0.0] for k in fam}
for n, (cnt, kb) in F.items():
    wkb = W.get(n, [cnt, 0.0])[1]
    lines.append('| `%s` | %.1f | %.2f | %.2f | %.2f |' % (n, cnt / steps, kb * 1.024 / cnt / 1e3, 2 * kb * 1.024 / cnt / 1e3, wkb * 1.024 / max(W.get(n, [cnt])[0], 1) / 1e3))
    tf += kb; tw += wkb
    for k, pats in fam.items():
        if any(p.lstrip('`') in n and not (p.startswith('`') and not n.startswith(p[1:])) for p in pats):
            famtot[k][0] += 2 * kb * 1024.0 / steps; famtot[k][1] += wkb * 1024.0 / steps
lines += ['', 'Whole step: FETCH raw %.2f GB (%.2f GB corrected), WRITE %.2f GB per step.' % (tf * 1.024 / steps / 1e6, 2 * tf * 1.024 / steps / 1e6, tw * 1.024 / steps / 1e6)]
open(out_md, 'w').write('\n'.join(lines) + '\n')
js = {}
for k, (fb, wb) in famtot.items():
    calls = sum(c for n, (c, _) in F.items() if main[k] in n) / steps      # operator calls per step = main-kernel launches
    if calls == 0:
        continue
    js[k] = dict(launches_per_step=calls, fetch_bytes_per_launch_corrected=fb / calls,
                 write_bytes_per_launch=wb / calls, hbm_bytes_per_launch=(fb + wb) / calls,
                 source='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py, FETCH x2 per MI355X_MICROARCH.md; '
                        'per conv operator call = main kernel + its split reduce')
json.dump(js, open(out_json, 'w'), indent=1)
print('\n'.join(lines[-3:])); print(json.dumps(js, indent=1))
