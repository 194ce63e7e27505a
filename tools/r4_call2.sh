#!/bin/bash
# round 4: the queue form in the net -- bit-identity tests, then same-box A/B of the form and its knobs
cd "$(dirname "$0")/.."; out=gpurun_out/r4_4; mkdir -p $out
(timeout 900 python -m pytest tests/test_queue.py -m gpu -q -x > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log) < /dev/null
tail -n 5 $out/pytest.log | cut -c1-300
tools/envsweep.sh $out/sweep "HRN_QUEUE=0" "HRN_QUEUE=1" "HRN_Q_TPB=2" "HRN_Q_BBF_SCALE=0.85" "HRN_Q_BBF_SCALE=1.15" "HRN_QUEUE=0" "HRN_QUEUE=1" "HRN_Q_TPB=2" "HRN_Q_BBF_SCALE=0.7" "HRN_Q_BBF_SCALE=1.3"
