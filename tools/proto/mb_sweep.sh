#!/bin/bash
# per-shape time tables at micro-batch 256 / 64 / 32 (tools/layer_profile.py), scaled to 256 crops: where smaller tensors (Infinity-Cache
# resident) win and where they lose
out=gpurun_out/mb_sweep; mkdir -p $out
for mb in 256 64 32; do
  timeout 100 python tools/layer_profile.py --mb $mb --reps 3 > $out/lp$mb.txt 2>&1 < /dev/null
  head -2 $out/lp$mb.txt | tail -1 | cut -c1-200
done
