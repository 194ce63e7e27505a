#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4_16; mkdir -p $out
timeout 600 python -m pytest tests/test_stem_fused.py -m gpu -x -q > $out/tests.txt 2>&1 < /dev/null; echo "tests rc=$?"; tail -3 $out/tests.txt
HRN_LIB_TAG=sftime timeout 300 python tools/debug/stemf_timing.py > $out/timing.txt 2>&1 < /dev/null
cat $out/timing.txt | tail -14
bash tools/envsweep.sh $out/ab "" "HRN_DISABLE_STEM_FUSE=1" 2>&1 | tee $out/ab.txt
