#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4_15; mkdir -p $out
timeout 300 python tools/debug/stemf_probe.py 64 64 1 > $out/probe64.txt 2>&1 < /dev/null
timeout 300 python tools/debug/stemf_probe.py 384 288 1 > $out/probe384.txt 2>&1 < /dev/null
head -60 $out/probe64.txt
