#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4_9; mkdir -p $out
timeout 200 tools/bin/c3q_test 256 > $out/c3q.txt 2>&1; echo rc=$?; grep -v "first bad" $out/c3q.txt | tail -n 16
(timeout 600 python -m pytest tests/test_queue.py -m gpu -q -x > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log) < /dev/null
tail -n 3 $out/pytest.log | cut -c1-300
tools/envsweep.sh $out/sweep "HRN_QUEUE=0" "HRN_QUEUE=1" "HRN_Q_BBF_SCALE=1.15" "HRN_Q_BBF_SCALE=1.3" "HRN_QUEUE=0" "HRN_QUEUE=1" "HRN_Q_BBF_SCALE=1.15" "HRN_Q_BBF_SCALE=1.3"
