#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4_17; mkdir -p $out
timeout 300 python tools/conv_table.py > $out/conv_table.txt 2>&1 < /dev/null
cat $out/conv_table.txt | tail -60
