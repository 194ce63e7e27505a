#!/bin/bash
out=gpurun_out/r6_exp5; mkdir -p $out
NO_BUILD=1 timeout 300 python tools/cu_timeline.py 2>&1 | grep -v amdgpu.ids | tee $out/cu_timeline.txt | cut -c1-400
