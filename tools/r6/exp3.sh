#!/bin/bash
# round 6, GPU call 3: where a small call (8 crops, the per-frame live loop) spends its time, per kernel; launch times of the graded launches by type
out=gpurun_out/r6_exp3; mkdir -p $out; R=$PWD; export TMPDIR=/tmp
timeout 300 python tools/clip_trace.py 8 > $out/clip_trace_8.txt 2>&1; cat $out/clip_trace_8.txt | tail -n 5
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$out/stats -o s --output-format csv -- bash -c "cd $R && python tools/clip_trace.py 8" > /dev/null 2>&1 < /dev/null)
cp "$(find $out/stats -name '*kernel_stats.csv' | head -1)" $out/clip8_kernel_stats.csv; rm -rf $out/stats
head -n 30 $out/clip8_kernel_stats.csv | cut -c1-200
timeout 200 python tools/launch_times.py > $out/launch_times.txt 2>&1; tail -n 30 $out/launch_times.txt | cut -c1-200
timeout 200 python tools/conv_table.py 48 384 288 8 > $out/conv_table_8.txt 2>&1; tail -n 40 $out/conv_table_8.txt | cut -c1-200
