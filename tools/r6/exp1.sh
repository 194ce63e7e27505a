#!/bin/bash
# round 6, GPU call 1: the three experiments of VERDICT r5 item 1 + a same-box baseline of the pass
#  (a) bench.py random vs --zeros (same box, interleaved)   (b) mfma_ceiling on 256 and on 64 CUs, contiguous and strided accesses
#  (c) the phase counters of the shipped kernel; static priority off (A/B)
out=gpurun_out/r6_exp1; mkdir -p $out
export HRN_DEBUG_ENV=1
B="--steps 8 --warmup 2 --no-clip --no-config1 --no-fp32-w48 --no-two-lanes --no-prepath --no-cpu-baseline"
for rep in 1 2; do
  timeout 200 python bench.py $B > $out/random_$rep.json 2> $out/random_$rep.err < /dev/null; python tools/abline.py random$rep $out/random_$rep.json
  timeout 200 python bench.py $B --zeros > $out/zeros_$rep.json 2> $out/zeros_$rep.err < /dev/null; python tools/abline.py zeros$rep $out/zeros_$rep.json
done
tools/ab.sh $out noprio default noprio
timeout 300 tools/bin/mfma_ceiling 256 1 > $out/ceiling_256.txt 2>&1
timeout 300 tools/bin/mfma_ceiling 64 1 > $out/ceiling_64.txt 2>&1
timeout 300 tools/bin/mfma_ceiling 128 1 > $out/ceiling_128.txt 2>&1
cat $out/ceiling_256.txt | cut -c1-230 | tail -n 34
echo ---- 64; cat $out/ceiling_64.txt | cut -c1-230 | tail -n 34
NO_BUILD=1 MB=256 timeout 300 python tools/c3_timing.py > $out/phase_timing.txt 2>&1
head -n 50 $out/phase_timing.txt | cut -c1-250
timeout 200 python tools/conv_table.py > $out/conv_table.txt 2>&1; tail -n 60 $out/conv_table.txt | cut -c1-200
