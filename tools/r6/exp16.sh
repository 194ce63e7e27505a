#!/bin/bash
# round 6, GPU call 16: stage-loop streams with the weight pieces walking a realistic footprint (shared by all blocks of a cout tile) instead of the same 24 KiB every stage
mkdir -p gpurun_out/r6_exp16
timeout 300 tools/bin/mfma_ceiling 256 2 > gpurun_out/r6_exp16/ceiling_256_w.txt 2>&1; cut -c1-220 gpurun_out/r6_exp16/ceiling_256_w.txt
