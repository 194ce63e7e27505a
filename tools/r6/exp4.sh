#!/bin/bash
# round 6, GPU call 4: TIMING-ONLY experiment -- the 96-cout form with every slab piece / residual load / store shaped as in a [C/32][rows][32]
# layout (contiguous KiB per instruction; results are garbage) against the shipped NHWC accesses (sixteen 64-byte segments per instruction)
out=gpurun_out/r6_exp4; mkdir -p $out
export HRN_DEBUG_ENV=1
for rep in 1 2; do
  for tag in default fakeblk; do
    if [ $tag = default ]; then unset HRN_LIB_TAG; else export HRN_LIB_TAG=$tag; fi
    echo "== $tag $rep"; timeout 200 python tools/launch_times.py 2>&1 | grep -v amdgpu.ids | tee -a $out/launch_times_$tag.txt
  done
done
unset HRN_LIB_TAG
tools/ab.sh $out default fakeblk default fakeblk
