#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4_26; mkdir -p $out
for tag in "" perfrag; do HRN_LIB_TAG=$tag timeout 200 python tools/ab_bits.py 2>/dev/null | tail -3; done
for tag in "" perfrag "" perfrag "" perfrag; do
  HRN_LIB_TAG=$tag timeout 120 python bench.py --steps 8 --warmup 2 --lanes 1 --no-clip --no-config1 --no-fp32-w48 --no-two-lanes --no-prepath --no-cpu-baseline --no-peaked > $out/b_$tag.json 2> $out/b_$tag.err < /dev/null
  python tools/abline.py "[$tag]" $out/b_$tag.json < /dev/null | cut -c1-130
done
