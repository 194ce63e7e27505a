#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4_11; mkdir -p $out
(timeout 900 python -m pytest tests/test_multi_gpu.py tests/test_round3.py tests/test_queue.py -m gpu -q -x -rs > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log) < /dev/null
tail -n 8 $out/pytest.log | cut -c1-400
timeout 300 python tools/layer_profile.py > $out/layers.txt 2>&1; tail -n 60 $out/layers.txt
