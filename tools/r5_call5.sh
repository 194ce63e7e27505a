#!/bin/bash
out=gpurun_out/r5_5; mkdir -p $out
timeout 300 tools/bin/mfma_ceiling > $out/mfma_ceiling.txt 2>&1 < /dev/null
grep -E "random" $out/mfma_ceiling.txt | cut -c1-230
