#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4_32; mkdir -p $out
bash tools/envsweep.sh $out/sw "" "HRN_BBF_TPB_DIV=2" "HRN_BBF_TPB_DIV=4" "HRN_BBF_TPB_DIV=6" "" "HRN_COMPACT_MIN_PAD=0.04" "HRN_LONG_SHARE=0.7" "" 2>&1 | cut -c1-150
