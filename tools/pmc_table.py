import csv, sys, collections
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    agg = collections.OrderedDict()
    for r in rows:
        k = r['Kernel_Name'][:60]
        d = agg.setdefault(k, collections.OrderedDict())
        d.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    for k, d in agg.items():
        print(k)
        for c, v in d.items():
            print('   %-34s n=%d  mean=%.4g' % (c, len(v), sum(v) / len(v)))
