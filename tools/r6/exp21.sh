#!/bin/bash
# round 6, GPU call 21: fuse kernel with multiply-shift index arithmetic -- parity tests, then same-box A/B against the 64-bit divisions
out=gpurun_out/r6_exp21; mkdir -p $out
export HRN_DEBUG_ENV=1
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_poseresnet.py tests/test_round2_gpu.py -m gpu -x -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log); tail -n 4 $out/tests.log | cut -c1-300
unset HRN_DEBUG_ENV
tools/ab.sh $out default fuseslow default fuseslow default fuseslow
