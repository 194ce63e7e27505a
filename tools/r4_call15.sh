#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4_22; mkdir -p $out
HRN_LIB_TAG=s2time timeout 300 python tools/debug/s2_timing.py > $out/s2_timing.txt 2>&1 < /dev/null; tail -4 $out/s2_timing.txt
