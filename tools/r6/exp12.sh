#!/bin/bash
# round 6, GPU call 12: the two settings of the block-length sweep that were above the defaults, three interleaved pairs each; the new GPU tests
(HRN_DEBUG_ENV=1 timeout 900 python -m pytest tests/test_xl.py tests/test_gpu_parity.py -m gpu -x -q -k "small_launch or scheduling" > gpurun_out/r6_exp12_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6_exp12_tests.log); tail -n 3 gpurun_out/r6_exp12_tests.log
tools/envsweep.sh gpurun_out/r6_exp12 "" "HRN_LONG_FACTOR=8" "HRN_LONG_FACTOR=8 HRN_BBF_TPB_DIV=1" "" "HRN_LONG_FACTOR=8" "HRN_LONG_FACTOR=8 HRN_BBF_TPB_DIV=1" "" "HRN_LONG_FACTOR=8" "HRN_LONG_FACTOR=8 HRN_BBF_TPB_DIV=1"
