import csv, collections, sys, glob
f = glob.glob(sys.argv[1] + '/*counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'gemm' in k or 'msda' in k or 'ffn' in k or 'layer' in k:
        name = k.replace('void ddp::', '').replace('ddp::(anonymous namespace)::', '')[:48]
        print(name, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
