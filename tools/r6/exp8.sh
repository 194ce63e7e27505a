#!/bin/bash
out=gpurun_out/r6_exp8; mkdir -p $out
NO_BUILD=1 MB=8 STEP=1 CLASSES=30 timeout 300 python tools/cu_timeline.py 2>&1 | grep -v amdgpu.ids | tee $out/cu_timeline_8.txt | cut -c1-330
