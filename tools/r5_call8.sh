#!/bin/bash
out=gpurun_out/r5_8; mkdir -p $out
for tag in "" w12 rf w12rf "" w12 rf w12rf; do
  echo "== [$tag]"; HRN_LIB_TAG=$tag timeout 120 python tools/layer1_time.py 2>&1 | grep "total" | tee -a $out/layer1.txt
done
tools/ab.sh $out/ab default w12 rf w12rf default w12 rf w12rf 2>&1 | tee $out/ab.txt
