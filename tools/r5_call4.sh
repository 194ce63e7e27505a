#!/bin/bash
out=gpurun_out/r5_4; mkdir -p $out
(timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "predict_stream" > $out/tests.log 2>&1; echo "rc=$?" >> $out/tests.log) < /dev/null
tail -n 5 $out/tests.log
for sdma in "" "HSA_ENABLE_SDMA=0"; do
  echo "== [$sdma]"
  env $sdma timeout 200 python tools/pcie_diag.py 2>&1 | grep "^{" | tee -a $out/pcie.txt
done
(timeout 300 python -m pytest tests/test_multi_gpu.py -m gpu -x -q > $out/tests_mg.log 2>&1; echo "rc=$?" >> $out/tests_mg.log) < /dev/null
tail -n 5 $out/tests_mg.log
timeout 300 tools/bin/mfma_ceiling > $out/mfma_ceiling.txt 2>&1 < /dev/null
grep -E "^T |^H " $out/mfma_ceiling.txt | cut -c1-230
