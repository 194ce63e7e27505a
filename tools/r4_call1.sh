#!/bin/bash
# round 4, first GPU call: the new tests, smoke(), the bench line with its new legs, the 32x32x16 side experiment
cd "$(dirname "$0")/.."; out=gpurun_out/r4_1; mkdir -p $out
(timeout 600 python -m pytest tests/test_peaked.py tests/test_s2.py tests/test_round3.py -m gpu -q -rP -x > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log) < /dev/null
grep -E "peaked|passed|failed|rc=" $out/pytest.log | cut -c1-400
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "rc=$?" >> $out/smoke.log) < /dev/null
tail -n 4 $out/smoke.log | cut -c1-300
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err < /dev/null
python - <<'PY'
import json
j = json.load(open("gpurun_out/r4_1/bench.json"))
print("value", j["value"], "frac", j["roofline"]["frac"], "ms", j["ms_per_step"])
print("peaked", json.dumps(j["parity"].get("peaked"))[:1500])
print("fp32_w48", json.dumps(j.get("fp32_w48_384x288"))[:600])
print("parity", {k: v for k, v in j["parity"].items() if "frac" in k or "identical" in k})
PY
timeout 120 tools/bin/mfma32_loop > $out/mfma32.txt 2>&1; cat $out/mfma32.txt
