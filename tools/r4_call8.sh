#!/bin/bash
# round 4, call 8: the fused stem kernel -- bit-identity tests, then a same-box A/B and its kernel times
cd "$(dirname "$0")/.."; R=$PWD; out=gpurun_out/r4_14; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_stem_fused.py -m gpu -x -q > $out/tests.txt 2>&1 < /dev/null; echo "tests rc=$?" | tee -a $out/tests.txt
tail -5 $out/tests.txt
bash tools/envsweep.sh $out/ab "" "HRN_DISABLE_STEM_FUSE=1" "" "HRN_DISABLE_STEM_FUSE=1" 2>&1 | tee $out/ab.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/$out/kt" -o k --output-format csv -- bash -c "cd $R && python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-clip --no-config1 --no-fp32-w48 --no-prepath --no-peaked" > /dev/null 2>&1 < /dev/null)
python - "$(find $out/kt -name '*kernel_stats.csv' | head -1)" <<'PY' | tee $out/stats.txt
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-90s calls %5s  avg %9.1f us  %5s %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
rm -rf $out/kt
