#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4_24; mkdir -p $out
for l in 1 2 1 2 1 2 3; do
  timeout 120 python bench.py --steps 10 --warmup 2 --lanes $l --no-clip --no-config1 --no-fp32-w48 --no-prepath --no-cpu-baseline --no-peaked > $out/l$l.json 2> $out/l$l.err < /dev/null
  python tools/abline.py "lanes=$l" $out/l$l.json < /dev/null | cut -c1-120
done
