#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4_21; mkdir -p $out
for tag in "" s2d2 "" s2d2; do
  HRN_LIB_TAG=$tag timeout 200 python tools/conv_table.py > $out/t_$tag.txt 2>&1 < /dev/null
  echo "[$tag] slab rows: $(grep '^slab' $out/t_$tag.txt | awk '{s+=$7} END {print s}') ms   $(head -2 $out/t_$tag.txt | tail -1 | cut -c1-120)"
done
HRN_LIB_TAG=s2d2 timeout 300 python -m pytest tests/test_s2.py -m gpu -x -q 2>&1 | tail -2
