import csv, glob, collections, sys
tag = sys.argv[1]
for i in (1, 2, 3, 4, 5):
    fs = glob.glob('gpurun_out/%s_pmc%d/*counter_collection.csv' % (tag, i))
    if not fs:
        print(i, 'no file'); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(fs[0])):
        agg[row['Kernel_Name'][:44]][row['Counter_Name']] += float(row['Counter_Value'])
    for k, d in agg.items():
        if 'conv' in k: print(i, k, {c: round(v / 1e6, 1) for c, v in d.items()})
    tr = glob.glob('gpurun_out/%s_pmc%d/*kernel_trace.csv' % (tag, i))
    if i == 3 and tr:
        t = collections.defaultdict(list)
        for r in csv.DictReader(open(tr[0])):
            t[r['Kernel_Name'][:44]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        for k, v in t.items():
            if 'conv' in k: print('   us', k, [round(x) for x in v], 'VGPR/AGPR/LDS', )
