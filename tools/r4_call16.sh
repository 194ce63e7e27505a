#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4_23; mkdir -p $out
bash tools/envsweep.sh $out/sw "" "HRN_S2_BLOCKS=512" "HRN_CHAIN_BLOCKS=256" "HRN_CHAIN_BLOCKS=768" "" "HRN_LONG_SHARE=0.9" "HRN_LONG_SHARE=0.75" "HRN_HALF_STAGES=6" "HRN_HALF_STAGES=12" "" "HRN_LONG_FACTOR=2" "HRN_LONG_FACTOR=6" 2>&1 | tee $out/sweep.txt
