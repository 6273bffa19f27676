"""Per-kernel totals of the LAST (event-instrumented, single-stream) step in a rocprofv3 kernel trace CSV."""
import csv, re, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows[a + 1:b + 1]:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); n = re.sub(r'\(.*', '', n).replace('void ', '')
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg[n][0] += 1; agg[n][1] += d
tot = sum(v[1] for v in agg.values())
span = (int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3
print('sum %.1f us, span %.1f us' % (tot, span))
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-55s %4d %9.1f us %5.1f%%' % (n[:55], c, d, 100 * d / tot))
