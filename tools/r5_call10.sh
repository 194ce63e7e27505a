#!/bin/bash
out=gpurun_out/r5_10; mkdir -p $out
NO_BUILD=1 MB=256 timeout 300 python tools/c3_timing.py > $out/c3_timing.txt 2>&1 < /dev/null
grep -v amdgpu.ids $out/c3_timing.txt | cut -c1-220 | head -60
