#!/bin/bash
# round 6, GPU call 20: the stride-2 slab kernel for cin = 96 (one 16-cout fragment per wave) -- bit identity, per-op pin, then same-box A/B by HRN_S2_CIN96
out=gpurun_out/r6_exp20; mkdir -p $out
export HRN_DEBUG_ENV=1
(timeout 1500 python -m pytest tests/test_s2.py tests/test_xl.py tests/test_bf16_pin.py -m gpu -x -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log); tail -n 12 $out/tests.log | cut -c1-300
B="--steps 8 --warmup 2 --no-clip --no-config1 --no-fp32-w48 --no-two-lanes --no-prepath --no-cpu-baseline"
for rep in 1 2 3; do
  HRN_S2_CIN96=0 timeout 200 python bench.py $B > $out/off_$rep.json 2> $out/off_$rep.err < /dev/null; python tools/abline.py s96_off$rep $out/off_$rep.json
  timeout 200 python bench.py $B > $out/on_$rep.json 2> $out/on_$rep.err < /dev/null; python tools/abline.py s96_on$rep $out/on_$rep.json
done
HRN_S2_CIN96=0 timeout 200 python tools/conv_table.py 2>&1 | grep -E "s2" > $out/conv_table_off.txt; cat $out/conv_table_off.txt
timeout 200 python tools/conv_table.py 2>&1 | grep -E "s2" > $out/conv_table_on.txt; cat $out/conv_table_on.txt
