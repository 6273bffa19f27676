"""Count instruction classes per basic block of a kernel in hipcc -S output (VALU steals fp32-MFMA cycles)."""
import re, collections, sys
path, key = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 50
s = open(path).read()
m = re.search(r'^(_Z\w*%s\w*):' % key, s, re.M)
body = s[m.end():]
body = body[:body.index('s_endpgm')]
blocks = []; cur = []; name = 'entry'
for l in body.split('\n'):
    l = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append((name, cur)); name = l; cur = []
    elif l and not l.startswith(';') and not l.startswith('.'):
        cur.append(l.split()[0])
blocks.append((name, cur))
for n, b in blocks:
    if len(b) < minlen: continue
    c = collections.Counter()
    for op in b:
        if op.startswith('v_mfma'): c['mfma'] += 1
        elif op.startswith('v_accvgpr'): c['accvgpr'] += 1
        elif op.startswith('v_'): c['valu'] += 1
        elif op.startswith('ds_'): c['lds'] += 1
        elif op.split('_')[0] in ('buffer', 'global', 'scratch', 'flat'): c['vmem:' + op.split('_')[0]] += 1
        elif op.startswith('s_waitcnt'): c['wait'] += 1
        elif op.startswith('s_'): c['salu'] += 1
    print(n, len(b), dict(c))
    vc = collections.Counter(op for op in b if op.startswith('v_') and not op.startswith('v_mfma'))
    print('    ', vc.most_common(14))
# per-MFMA-slot distribution of VALU instructions in blocks containing MFMAs
for n, b in blocks:
    if not any(op.startswith('v_mfma') for op in b): continue
    segs = []; cur = 0
    for op in b:
        if op.startswith('v_mfma'): segs.append(cur); cur = 0
        elif op.startswith('v_') and not op.startswith('v_accvgpr'): cur += 1
    segs.append(cur)
    print('VALU between MFMAs in', n.split()[0], ':', segs, ' slots with VALU:', sum(1 for x in segs if x))
