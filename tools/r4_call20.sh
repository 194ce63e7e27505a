#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4_28; mkdir -p $out
for s in "" "HRN_DISABLE_COMPACT=1" "" "HRN_DISABLE_COMPACT=1"; do
  env $s timeout 200 python tools/conv_table.py > $out/t.txt 2>&1 < /dev/null
  echo "[$s] $(grep -E '^(n96|fused-bb)' $out/t.txt | awk '{printf "%s-%s: %.3f ms %d TF | ", $4, $6, $8, $10}')"
done
