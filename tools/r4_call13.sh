#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4_20; mkdir -p $out
HRN_LIB_TAG=sftime timeout 300 python tools/debug/stemf_timing.py > $out/timing.txt 2>&1 < /dev/null
tail -16 $out/timing.txt
(rocm-smi --showclocks --showpower 2>/dev/null | head -30) > $out/smi_idle.txt; tail -12 $out/smi_idle.txt
( python bench.py --steps 60 --warmup 2 --no-clip --no-config1 --no-fp32-w48 --no-prepath --no-cpu-baseline --no-peaked > $out/b.json 2> $out/b.err < /dev/null & )
sleep 14; (rocm-smi --showclocks --showpower 2>/dev/null | head -30) > $out/smi_load.txt; sleep 1; (rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power") >> $out/smi_load.txt
wait; sleep 8
grep -i "sclk\|power\|mclk\|fclk" $out/smi_load.txt | head -12
