#!/bin/bash
# end-of-round evidence in one bounded call on the GPU box: GPU tests, the bench line, rocprofv3 kernel stats of the same
# command, the PMC traffic passes, and one wave-state / matrix-pipe PMC pass.   tools/final_run.sh <round> <outdir under gpurun_out>
round=${1:-2}; out=gpurun_out/${2:-final_r$round}
cd "$(dirname "$0")/.."; R=$PWD; export TMPDIR=/tmp
mkdir -p "$out"
(timeout 900 python -m pytest tests -m gpu -x -q > "$out/gputest.log" 2>&1; echo "rc=$?" >> "$out/gputest.log") < /dev/null
tail -n 3 "$out/gputest.log"
timeout 400 python bench.py > "$out/bench.json" 2> "$out/bench.err" < /dev/null
python tools/abline.py bench "$out/bench.json" < /dev/null
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$R/$out/stats" -o r$round --output-format csv -- bash -c "cd $R && python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-clip --no-config1 --no-prepath > $R/$out/bench_under_rocprof.json 2>/dev/null" > /dev/null 2>&1 < /dev/null)
cp "$(find $out/stats -name '*kernel_stats.csv' | head -1)" "$out/round${round}_kernel_stats.csv" 2>/dev/null
head -n 12 "$out/round${round}_kernel_stats.csv" | cut -c1-160
timeout 500 tools/pmc_traffic.sh $round < /dev/null | cut -c1-400
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d "$R/$out/pmc_wave" -o w --output-format csv -- bash -c "cd $R && python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-clip --no-config1 --no-prepath" > /dev/null 2>&1 < /dev/null)
python tools/pmc_mean.py "$(find $out/pmc_wave -name '*counter_collection.csv' | head -1)" "conv3x3_lds_kernel<48" > "$out/round${round}_pmc_wave.txt" < /dev/null
cat "$out/round${round}_pmc_wave.txt"
rm -rf "$out/stats" "$out/pmc_wave"
