#!/bin/bash
out=gpurun_out/ab_loader2; mkdir -p $out
i=0
for v in ${ARMS:-0 12 6 0 12}; do
  i=$((i+1))
  HRN_N96_LOADER=$v NO_BUILD=1 timeout 120 python tools/build_variant.py loader "" --steps 8 --warmup 2 --no-clip --no-config1 --no-prepath --no-cpu-baseline > $out/$i.json 2> $out/$i.err < /dev/null
  python tools/abline.py "[loader>=$v]" $out/$i.json < /dev/null
done
