import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# group consecutive dispatches by kernel name in order; print avg duration per (name, grid)
agg = collections.OrderedDict()
for r in rows:
    name = r['Kernel_Name']
    if 'bn_' not in name: continue
    key = (name[:60], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg.setdefault(key, []).append(d)
for k, v in agg.items():
    v = v[3:]
    print("%-62s grid %-9s n=%3d avg %7.1f us min %7.1f" % (k[0], k[1], len(v), sum(v) / max(1, len(v)), min(v) if v else 0))
