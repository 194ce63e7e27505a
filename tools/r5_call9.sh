#!/bin/bash
out=gpurun_out/r5_9; mkdir -p $out
(timeout 1500 python -m pytest tests -m gpu -q -x > $out/gputest.log 2>&1; echo "rc=$?" >> $out/gputest.log) < /dev/null
tail -n 25 $out/gputest.log | cut -c1-400
