#!/bin/bash
# round 6, GPU call 24: chain kernel -- rolling prefetch of the next fragment's 18 tap fragments + three-operation epilogues: bit identity, layer1 times, bench
out=gpurun_out/r6_exp24; mkdir -p $out
export HRN_DEBUG_ENV=1
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_bf16_pin.py tests/test_poseresnet.py -m gpu -x -q > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log); tail -n 6 $out/tests.log | cut -c1-300
unset HRN_DEBUG_ENV
timeout 200 python tools/layer1_time.py 2>&1 | grep -v amdgpu.ids | tee $out/layer1.txt
B="--steps 8 --warmup 2 --no-clip --no-config1 --no-fp32-w48 --no-two-lanes --no-prepath --no-cpu-baseline"
for rep in 1 2 3; do timeout 200 python bench.py $B > $out/b_$rep.json 2> $out/b_$rep.err < /dev/null; python tools/abline.py new$rep $out/b_$rep.json; done
