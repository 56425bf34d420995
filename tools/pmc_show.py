import csv,glob,collections,sys
for d in sys.argv[1:]:
    f=glob.glob(f'gpurun_out/{d}/*/*counter_collection.csv')[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void gear_hip::','')
        if 'gear_hip' in r['Kernel_Name']:
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(d, k, {c: round(sum(x)/len(x)) for c,x in sorted(v.items())})
