#!/bin/bash
# round 6, GPU call 6: the XL form of the generic kernel (stride-2 3x3, cin % 32 == 0): bit identity, then same-box A/B by the creation-time switch
out=gpurun_out/r6_exp6; mkdir -p $out
export HRN_DEBUG_ENV=1
(timeout 900 python -m pytest tests/test_xl.py -m gpu -x -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log); tail -n 15 $out/tests.log | cut -c1-300
B="--steps 8 --warmup 2 --no-clip --no-config1 --no-fp32-w48 --no-two-lanes --no-prepath --no-cpu-baseline"
for rep in 1 2 3; do
  HRN_DIRECT_XLDS=0 timeout 200 python bench.py $B > $out/off_$rep.json 2> $out/off_$rep.err < /dev/null; python tools/abline.py xl_off$rep $out/off_$rep.json
  timeout 200 python bench.py $B > $out/on_$rep.json 2> $out/on_$rep.err < /dev/null; python tools/abline.py xl_on$rep $out/on_$rep.json
done
HRN_DIRECT_XLDS=0 timeout 200 python tools/conv_table.py 2>&1 | grep -E "s2|kernel" | grep -E "generic|kernel" > $out/conv_table_off.txt; cat $out/conv_table_off.txt
timeout 200 python tools/conv_table.py 2>&1 | grep -E "s2|kernel" | grep -E "generic|kernel" > $out/conv_table_on.txt; cat $out/conv_table_on.txt
