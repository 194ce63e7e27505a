#!/bin/bash
out=gpurun_out/r5_2; mkdir -p $out
timeout 400 python tools/lane_stagger.py 8 > $out/stagger.txt 2>&1 < /dev/null
cat $out/stagger.txt | grep -v amdgpu.ids
