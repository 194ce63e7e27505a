#!/bin/bash
cd "$(dirname "$0")/.."; R=$PWD; out=gpurun_out/r4_18; mkdir -p $out; export TMPDIR=/tmp
timeout 300 python tools/clip_trace.py > $out/clip.txt 2>&1 < /dev/null; tail -5 $out/clip.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$R/$out/kt" -o k --output-format csv -- bash -c "cd $R && python tools/clip_trace.py" > /dev/null 2>&1 < /dev/null)
python - "$(find $out/kt -name '*kernel_trace.csv' | head -1)" <<'PY' | tee $out/trace.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last timed clip run = the 30 frames before the 'forward alone' loop: find stem kernels (one per pass)
stems = [i for i, r in enumerate(rows) if "stem_fused_kernel" in r["Kernel_Name"] or "stem_mfma" in r["Kernel_Name"]]
print("passes in trace:", len(stems))
# frames 61..90 are the third clip run (2 warm-ups of 30)
a, b = stems[60], stems[90] if len(stems) > 90 else len(rows)
seg = rows[a - 3:b - 3]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e6
wall = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
print("third clip run: %d kernels, GPU busy %.2f ms of %.2f ms wall (%.1f %%), %.1f kernels per frame" % (len(seg), busy, wall, 100 * busy / wall, len(seg) / 30))
gaps = sorted((int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e3 for x, y in zip(seg, seg[1:]))
print("gaps between kernels (us): median %.2f  p90 %.2f  max %.1f  sum %.2f ms" % (gaps[len(gaps) // 2], gaps[int(len(gaps) * 0.9)], gaps[-1], sum(gaps) / 1e3))
from collections import defaultdict
d = defaultdict(lambda: [0, 0.0])
for r in seg:
    k = r["Kernel_Name"].split("(")[0][:70]
    d[k][0] += 1; d[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-72s %5d  %8.1f us per frame  avg %6.1f us" % (k, v[0], v[1] / 30, v[1] / v[0]))
PY
rm -rf $out/kt
