#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4_27; mkdir -p $out
timeout 900 python -m pytest tests/test_compact.py -m gpu -x -q > $out/tests.txt 2>&1 < /dev/null; echo "tests rc=$?"; tail -12 $out/tests.txt | cut -c1-200
bash tools/envsweep.sh $out/ab "" "HRN_DISABLE_COMPACT=1" "" "HRN_DISABLE_COMPACT=1" 2>&1 | cut -c1-140
