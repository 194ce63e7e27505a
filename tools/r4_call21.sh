#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4_29; mkdir -p $out
for s in "" "HRN_DISABLE_COMPACT=1" "" "HRN_DISABLE_COMPACT=1"; do
  echo "== [$s]"; env $s timeout 200 python tools/launch_times.py 2>/dev/null | tail -6
done | tee $out/launch.txt
