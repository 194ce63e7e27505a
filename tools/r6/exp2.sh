#!/bin/bash
# round 6, GPU call 2: the 96-cout form without the no-residual loads -- element-wise tests, then same-box A/B against round 5's behaviour
out=gpurun_out/r6_exp2; mkdir -p $out
export HRN_DEBUG_ENV=1
(timeout 900 python -m pytest tests/test_n96.py tests/test_compact.py tests/test_bf16_pin.py tests/test_gpu_parity.py -m gpu -x -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log); tail -n 4 $out/tests.log
tools/ab.sh $out default resalways default resalways default resalways
timeout 300 tools/bin/mfma_ceiling 256 1 > $out/ceiling_256.txt 2>&1; cut -c1-200 $out/ceiling_256.txt | head -n 20
