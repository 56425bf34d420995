import collections, csv, glob, sys
for d in sorted(glob.glob(f"gpurun_out/{sys.argv[1]}/*/")):
    fs = glob.glob(d + "*/*counter_collection.csv")
    if not fs: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, m in agg.items():
        if 'k_sweep' not in k and 'k_narrow' not in k: continue
        w = sum(m["SQ_WAVES"]) / len(m["SQ_WAVES"])
        print(d.split('/')[-2], k, {c: round(sum(v) / len(v) / w, 1) for c, v in m.items() if c != 'SQ_WAVES'}, int(w))
