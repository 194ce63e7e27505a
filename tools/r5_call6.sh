#!/bin/bash
out=gpurun_out/r5_6; mkdir -p $out
for t in "" _b21 _b11 _b20 _b01; do
  echo "== c3n_test$t"
  timeout 200 tools/bin/c3n_test$t 256 > $out/c3n$t.txt 2>&1 < /dev/null; echo "rc=$?"
  grep -iE "mismatch|wrong|fail|bad|us per|TFLOP|ok" $out/c3n$t.txt | tail -n 14 | cut -c1-200
done
tools/ab.sh $out/ab default b21 b11 b20 b01 default b21 b11 b20 b01 default b21 2>&1 | tee $out/ab.txt
