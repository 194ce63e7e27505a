#!/bin/bash
# HBM-side bytes of the dominant kernel (bench.py: roofline.traffic): two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; counters
# only, --kernel-trace, as MI355X_MICROARCH.md's HBM section prescribes) of one 256-crop bench pass, reduced to
# profiles/round<N>_pmc_traffic.json together with the hash of the sources they were taken on.
#   tools/pmc_traffic.sh <round number>          (on the GPU box, from the repo root; ~1 minute)
set -e
round=${1:-2}
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp
out=gpurun_out/pmc_traffic_r$round
rm -rf "$out"; mkdir -p "$out"
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-prepath --no-clip --no-config1 --no-fp32-w48 --no-two-lanes"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$R/$out/fetch" -o f --output-format csv -- bash -c "cd $R && $B" > /dev/null 2>&1 < /dev/null)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$R/$out/write" -o w --output-format csv -- bash -c "cd $R && $B" > /dev/null 2>&1 < /dev/null)
# (written under gpurun_out/: that is what comes back from the GPU box; copy the three files into profiles/ afterwards)
python tools/pmc_traffic.py "$out/fetch" "$out/write" "$out/round${round}_pmc_traffic.json" < /dev/null
cp "$(find $out/fetch -name '*counter_collection.csv' | head -1)" "$out/round${round}_pmc_fetch_size.csv"
cp "$(find $out/write -name '*counter_collection.csv' | head -1)" "$out/round${round}_pmc_write_size.csv"
