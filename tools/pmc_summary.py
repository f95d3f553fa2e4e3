import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else r'gemm\d+_kernel(<\d, \d>)?'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    m = re.search(pat, r['Kernel_Name'])
    if not m: continue
    key = m.group(0) + " grid=" + r['Grid_Size']
    agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    agg[key]['_dur_us'].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
names = sorted({n for d in agg.values() for n in d})
for k in sorted(agg):
    print(k, "launches=%d" % len(agg[k]['_dur_us']))
    for n in names:
        v = agg[k][n]
        if v: print(f"    {n:30s} {sum(v)/len(v):16.4g}")
