"""Per-dispatch PMC table of one rocprofv3 --pmc run: python tools/pmc_dispatch.py <dir> <kernel substring> [max rows]."""
import csv, glob, collections, sys
d, pat = sys.argv[1], sys.argv[2]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 400
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
rows = collections.OrderedDict()
for row in csv.DictReader(open(f[0])):
    if pat not in row["Kernel_Name"]:
        continue
    key = int(row["Dispatch_Id"])
    r = rows.setdefault(key, {"kernel": row["Kernel_Name"].split("(")[0][-44:], "grid": row.get("Grid_Size", "?")})
    r[row["Counter_Name"]] = r.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for i, (k, r) in enumerate(rows.items()):
    if i >= lim:
        break
    print(k, r["kernel"], "grid", r["grid"], {c: round(v) for c, v in r.items() if c not in ("kernel", "grid")})
