"""HBM-side bytes per grouped BasicBlock launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
`bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-prepath`: writes profiles/<name>.json.
usage: python tools/pmc_traffic.py <fetch dir> <write dir> <out json> [launches per pass = 64] [stage-2 launches to skip = 8]"""
import collections, csv, glob, json, sys

fd, wd, out = sys.argv[1:4]
per_pass = int(sys.argv[4]) if len(sys.argv) > 4 else 64
skip = int(sys.argv[5]) if len(sys.argv) > 5 else 8
PAT = "conv3x3_lds_kernel<48, 3>"


def per_dispatch(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = collections.OrderedDict()
    for row in csv.DictReader(open(f)):
        if PAT in row["Kernel_Name"] and row["Counter_Name"] == counter:
            rows[int(row["Dispatch_Id"])] = rows.get(int(row["Dispatch_Id"]), 0.0) + float(row["Counter_Value"])
    vals = [v for _, v in sorted(rows.items())]
    assert len(vals) % per_pass == 0, (len(vals), per_pass)
    keep = [v for i, v in enumerate(vals) if i % per_pass >= skip]   # stage 3 / 4 only, as bench.py grades
    return keep


fetch, write = per_dispatch(fd, "FETCH_SIZE"), per_dispatch(wd, "WRITE_SIZE")
f_kb, w_kb = sum(fetch) / len(fetch), sum(write) / len(write)
res = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only), bench.py --steps 1 "
              "--warmup 1 --no-cpu-baseline --no-roofline --no-prepath, 256 crops",
    "kernel": "conv3x3_lds_kernel<48,3>, the stage-3/4 grouped launches (%d passes x %d)" % (len(fetch) // (per_pass - skip), per_pass - skip),
    "launches": len(fetch),
    "fetch_size_raw_kb_avg": round(f_kb, 1), "fetch_bytes_corrected_avg": round(f_kb * 1024 * 2),
    "write_size_raw_kb_avg": round(w_kb, 1), "write_bytes_avg": round(w_kb * 1024),
    "traffic_bytes_per_launch": round(f_kb * 1024 * 2 + w_kb * 1024),
    "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section: 128-B requests tallied at 64 B); WRITE_SIZE "
                  "uncorrected (uncalibrated); Infinity-Cache hits are included in both",
}
import importlib.util, os
_spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
_b = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_b)
res["source_hash"] = _b.source_hash()   # bench.py quotes the reading only on exactly these sources
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
