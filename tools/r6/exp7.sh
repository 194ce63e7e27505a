#!/bin/bash
# round 6, GPU call 7: small-call work (head slabs, per-conv tile size, grouped operand requests) -- the per-frame loop, then the full GPU suite
out=gpurun_out/r6_exp7; mkdir -p $out; R=$PWD; export TMPDIR=/tmp
timeout 300 python tools/clip_trace.py 8 > $out/clip_trace_8.txt 2>&1; tail -n 4 $out/clip_trace_8.txt
HRN_DEBUG_ENV=1 HRN_SMALL_KEEP=0 timeout 300 python tools/clip_trace.py 8 > $out/clip_trace_8_keep0.txt 2>&1; tail -n 4 $out/clip_trace_8_keep0.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$out/stats -o s --output-format csv -- bash -c "cd $R && python tools/clip_trace.py 8" > /dev/null 2>&1 < /dev/null)
cp "$(find $out/stats -name '*kernel_stats.csv' | head -1)" $out/clip8_kernel_stats.csv; rm -rf $out/stats
head -n 22 $out/clip8_kernel_stats.csv | cut -c1-200
timeout 200 python tools/latency.py > $out/latency.txt 2>&1; tail -n 8 $out/latency.txt
(timeout 1700 python -m pytest tests -m gpu -x -q > $out/gputest.log 2>&1; echo "rc=$?" >> $out/gputest.log); tail -n 6 $out/gputest.log | cut -c1-300
