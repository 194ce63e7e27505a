import csv, glob, collections, sys
d = sys.argv[1]
f = glob.glob(d + "/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"][:60]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k, dd in agg.items():
    if "hrn::" in k:
        print(k, {c: (round(v / cnt[k][c]), cnt[k][c]) for c, v in dd.items()})
