import csv, collections, glob, sys
for d in sys.argv[1:]:
    f = glob.glob(f'{d}/*/*_counter_collection.csv')[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'][:34], r['Counter_Name'])].append(float(r['Counter_Value']))
    ks = sorted({k for k,_ in agg})
    for k in ks:
        if 'whitted' not in k or 'true' in k: continue
        print(k, ' '.join(f"{c.replace('SQ_','')}={sum(v)/len(v):.4g}" for (kk,c),v in sorted(agg.items()) if kk==k))
