#!/bin/bash
out=gpurun_out/r5_7; mkdir -p $out
export HRN_DEBUG_ENV=1
(timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chain_kernel_is_bit_identical or scheduling_variants" > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log) < /dev/null
tail -n 15 $out/tests.log | cut -c1-300
for e in "" "HRN_DISABLE_CHAIN3=1" "" "HRN_DISABLE_CHAIN3=1"; do
  echo "== [$e]"; env $e timeout 120 python tools/layer1_time.py 2>&1 | grep -v amdgpu.ids | tee -a $out/layer1.txt
done
tools/envsweep.sh $out/sweep "" "HRN_DISABLE_CHAIN3=1" "" "HRN_DISABLE_CHAIN3=1" "" "HRN_DISABLE_CHAIN3=1" 2>&1 | tee $out/ab.txt
