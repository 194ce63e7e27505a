#!/bin/bash
# same-box sweep of creation-time switches with the shipped library: one bench line per setting
# usage: tools/envsweep.sh <outdir> "VAR=1 VAR2=3" "VAR=2" ...      ("" = defaults)
out=$1; shift
export HRN_DEBUG_ENV=1   # the library ignores HRN_* switches unless the process opts in
mkdir -p "$out"
i=0
for setting in "$@"; do
  i=$((i+1))
  env $setting timeout 120 python bench.py --steps 8 --warmup 2 --no-clip --no-config1 --no-fp32-w48 --no-two-lanes --no-prepath --no-cpu-baseline > "$out/$i.json" 2> "$out/$i.err" < /dev/null
  python tools/abline.py "[$setting]" "$out/$i.json" < /dev/null
done
