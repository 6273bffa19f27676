"""One step of a rocprofv3 --kernel-trace CSV as a per-queue timeline: per kernel start / duration / gap to the previous
kernel of the same queue.  usage: timeline.py <kernel_trace.csv> [step index from the end, default 3] [--full]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).replace('void ', '')
    r['n'] = re.sub(r'\(.*', '', n)[:44]
rows.sort(key=lambda r: r['s'])
adam = [i for i, r in enumerate(rows) if r['n'].startswith('adam_kernel')]
k = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
a0, a1 = adam[-k - 1], adam[-k]
step = rows[a0 + 1:a1 + 1]
t0 = rows[a0]['e']
print('step wall %.1f us (adam end -> adam end), %d kernels' % ((step[-1]['e'] - t0) / 1e3, len(step)))
last = {}
busy = []
for r in step:
    q = r['Queue_Id']
    gap = (r['s'] - last[q]) / 1e3 if q in last else 0.0
    last[q] = r['e']
    busy.append((r['s'], r['e']))
    if '--full' in sys.argv:
        print('q%s %9.1f +%7.1f gap %6.1f  grid %6d  %s' % (q, (r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, gap,
                                                     int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), r['n']))
# union of busy intervals, per-queue totals
busy.sort()
u, cs, ce = 0, None, None
for s, e in busy:
    if cs is None: cs, ce = s, e
    elif s <= ce: ce = max(ce, e)
    else: u += ce - cs; cs, ce = s, e
u += ce - cs
print('union of busy intervals %.1f us' % (u / 1e3))
for q in sorted(set(r['Queue_Id'] for r in step)):
    rs = [r for r in step if r['Queue_Id'] == q]
    print('queue %s: %d kernels, busy %.1f us, span %.1f .. %.1f' % (q, len(rs), sum(r['e'] - r['s'] for r in rs) / 1e3,
          (rs[0]['s'] - t0) / 1e3, (rs[-1]['e'] - t0) / 1e3))
