#!/bin/bash
# round 6, GPU call 25: chain kernel of this round against round 5's, same box: layer1 times and bench pairs
out=gpurun_out/r6_exp25; mkdir -p $out
for rep in 1 2; do
  for tag in chainold default; do
    if [ $tag = default ]; then unset HRN_LIB_TAG; else export HRN_LIB_TAG=$tag; fi
    echo "== $tag $rep: $(timeout 200 python tools/layer1_time.py 2>&1 | grep 'layer1 total')"
  done
done
unset HRN_LIB_TAG
tools/ab.sh $out chainold default chainold default chainold default
