"""Summarise a rocprofv3 rocpd SQLite database into a per-kernel table (like --stats CSV)."""
import sqlite3, sys, re
db = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in c.execute('pragma table_info(%s)' % kd)]
scol = [r[1] for r in c.execute('pragma table_info(%s)' % ks)]
namecol = 'kernel_name' if 'kernel_name' in scol else ('display_name' if 'display_name' in scol else 'name')
rows = c.execute('select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start' % (namecol, kd, ks)).fetchall()
agg = {}
tot = 0
for name, st, en in rows:
    name = re.sub(r'\(.*', '', name)
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    a = agg.setdefault(name, [0, 0])
    a[0] += 1; a[1] += en - st; tot += en - st
span = rows[-1][2] - rows[0][1]
print('total kernel time %.3f ms over %d dispatches; span %.3f ms' % (tot / 1e6, len(rows), span / 1e6))
print('%-90s %8s %12s %10s %6s' % ('kernel', 'calls', 'total_ms', 'avg_us', '%'))
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-90s %8d %12.3f %10.2f %6.2f' % (name[:90], n, t / 1e6, t / n / 1e3, 100.0 * t / tot))
