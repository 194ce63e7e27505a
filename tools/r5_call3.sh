#!/bin/bash
out=gpurun_out/r5_3; mkdir -p $out
timeout 300 tools/bin/mfma_ceiling > $out/mfma_ceiling.txt 2>&1 < /dev/null
cat $out/mfma_ceiling.txt
