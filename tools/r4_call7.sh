#!/bin/bash
cd "$(dirname "$0")/.."; R=$PWD; out=gpurun_out/r4_13; mkdir -p $out; export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$R/$out/kt" -o k --output-format csv -- bash -c "cd $R && python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-clip --no-config1 --no-fp32-w48 --no-prepath" > /dev/null 2>&1 < /dev/null)
CSV="$(find $out/kt -name '*kernel_trace.csv' | head -1)"
python - "$CSV" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
hot = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "conv3x3_lds_kernel<48" in r["Kernel_Name"]]
# the last pass: 64 launches (stage 2: 8, stage 3: 32, stage 4: 24), alternating conv1-type / conv2-type
last = hot[-64:]
def show(tag, xs):
    a, b = xs[0::2], xs[1::2]
    print("%s: conv1-type (with the fused 48-ch blocks) %.1f us, conv2-type %.1f us  (n = %d + %d)" % (tag, sum(a) / len(a), sum(b) / len(b), len(a), len(b)))
show("stage 2", last[:8]); show("stage 3", last[8:40]); show("stage 4", last[40:64])
print("stage 4 launches:", " ".join("%.0f" % x for x in last[40:64]))
# gaps between consecutive kernels of the last pass
allk = rows[-140:]
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(allk, allk[1:])]
print("gaps between consecutive kernels (us): mean %.2f  median %.2f  max %.2f  sum over the last 139: %.1f" % (sum(gaps) / len(gaps), sorted(gaps)[len(gaps) // 2], max(gaps), sum(gaps)))
PY
rm -rf $out/kt
