#!/bin/bash
# round 6, GPU call 9: MR == 1 operand ring of the 96-cout form -- element-wise tests, small-call timings, timeline at 8 crops
out=gpurun_out/r6_exp9; mkdir -p $out
export HRN_DEBUG_ENV=1
(timeout 900 python -m pytest tests/test_n96.py tests/test_compact.py tests/test_round2_gpu.py -m gpu -x -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log); tail -n 5 $out/tests.log | cut -c1-300
unset HRN_DEBUG_ENV
timeout 300 python tools/clip_trace.py 8 > $out/clip_trace_8.txt 2>&1; tail -n 4 $out/clip_trace_8.txt
timeout 200 python tools/latency.py > $out/latency.txt 2>&1; tail -n 3 $out/latency.txt
