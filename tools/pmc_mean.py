"""mean counter values per kernel (substring filter) of a rocprofv3 --pmc csv: python tools/pmc_mean.py <counter_collection.csv> [substr]"""
import collections, csv, sys
path, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "conv3x3_lds_kernel<48")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    if sub not in k:
        continue
    acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k[:60]][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, cs in acc.items():
    d = list(dur[k].values())
    print(k, "dispatches", len(d), "mean us %.1f" % (sum(d) / len(d)))
    for c, v in sorted(cs.items()):
        print("   %-40s mean %.4g" % (c, sum(v) / len(v)))
