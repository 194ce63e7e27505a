#!/bin/bash
# round 6, GPU call 11: block-length sweep of the grouped BasicBlock launches on the final code (creation-time switches, same box)
tools/envsweep.sh gpurun_out/r6_exp11 "" "HRN_BBF_TPB_DIV=2" "HRN_BBF_TPB_DIV=1" "HRN_LONG_FACTOR=8" "HRN_HALF_STAGES=12" "HRN_HALF_STAGES=16" "HRN_LONG_SHARE=0.92" "HRN_BBF_TPB_DIV=2 HRN_LONG_FACTOR=8" "" 
