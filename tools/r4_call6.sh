#!/bin/bash
cd "$(dirname "$0")/.."; R=$PWD; out=gpurun_out/r4_12; mkdir -p $out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_s2.py tests/test_bf16_pin.py -m gpu -q -x > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log) < /dev/null
tail -n 4 $out/pytest.log | cut -c1-300
timeout 300 python tools/layer_profile.py --mb 256 > $out/layers256.txt 2>&1; grep -E "pass of|s2 " $out/layers256.txt
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $PMC -d "$R/$out/pmc_wave" -o w --output-format csv -- bash -c "cd $R && python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-clip --no-config1 --no-fp32-w48 --no-prepath" > /dev/null 2>&1 < /dev/null)
CSV="$(find $out/pmc_wave -name '*counter_collection.csv' | head -1)"
python tools/pmc_mean.py "$CSV" "conv_s2_slab_kernel" > $out/round4_pmc_s2.txt < /dev/null
python tools/pmc_mean.py "$CSV" "conv3x3_lds_kernel<48" > $out/round4_pmc_wave.txt < /dev/null
cat $out/round4_pmc_s2.txt; rm -rf $out/pmc_wave
