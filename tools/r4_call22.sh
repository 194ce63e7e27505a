#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4_31; mkdir -p $out
for a in "256 192 250 256" "256 192 256 256" "256 192 64 256" "384 288 250 256"; do timeout 200 python tools/debug/compact_probe.py $a 2>&1 | grep -v amdgpu.ids | head -8; done | tee $out/probe.txt
