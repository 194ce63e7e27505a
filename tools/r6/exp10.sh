#!/bin/bash
out=gpurun_out/r6_exp10; mkdir -p $out
for rep in 1 2 3; do
  for tag in default nor3; do
    if [ $tag = default ]; then unset HRN_LIB_TAG; else export HRN_LIB_TAG=$tag; fi
    echo "== $tag $rep: $(timeout 300 python tools/clip_trace.py 8 2>&1 | grep -E 'per_frame|forward alone' | tr '\n' ' ')"
  done
done
unset HRN_LIB_TAG
