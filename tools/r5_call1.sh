#!/bin/bash
# round 5, GPU call 1: selected GPU tests on the pruned tree, the ceiling microbenchmark, prologue A/B, lanes pairs
out=gpurun_out/r5_1; mkdir -p $out
export HRN_DEBUG_ENV=0
(timeout 420 python -m pytest tests/test_compact.py tests/test_pads.py tests/test_n96.py tests/test_round3.py -m gpu -x -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log) < /dev/null
tail -n 4 $out/tests.log
timeout 200 tools/bin/mfma_ceiling > $out/mfma_ceiling.txt 2>&1 < /dev/null
cat $out/mfma_ceiling.txt
tools/ab.sh $out/ab default oldpro default oldpro default oldpro 2>&1 | tee $out/ab.txt
for i in 1 2 3; do
  timeout 200 python bench.py --steps 10 --warmup 2 --no-clip --no-config1 --no-fp32-w48 --no-prepath --no-cpu-baseline > $out/lanes$i.json 2> $out/lanes$i.err < /dev/null
  python - $out/lanes$i.json <<'PY'
import json,sys
d=[json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
print("value", d["value"], "frac", d["roofline"]["frac"], "two_lanes", d.get("two_lanes"))
PY
done 2>&1 | tee $out/lanes.txt
