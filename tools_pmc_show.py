import csv,glob,collections,sys
for d in sys.argv[1:]:
    f=glob.glob(f'gpurun_out/{d}/*/*counter_collection.csv')[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_check' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d, {c: round(sum(v)/len(v)/16384,1) for c,v in sorted(agg.items())})
