#!/bin/bash
# same-box A/B of prebuilt library variants (tools/build_variant.py NO_BUILD=1): one bench line per tag, key numbers only
# usage: tools/ab.sh <outdir> tag1 tag2 ...      (a tag "default" runs the shipped library)
out=$1; shift
mkdir -p "$out"
for tag in "$@"; do
  if [ "$tag" = default ]; then
    timeout 120 python bench.py --steps 8 --warmup 2 --no-clip --no-config1 --no-fp32-w48 --no-two-lanes --no-prepath --no-cpu-baseline > "$out/$tag.json" 2> "$out/$tag.err" < /dev/null
  else
    NO_BUILD=1 timeout 120 python tools/build_variant.py "$tag" "" --steps 8 --warmup 2 --no-clip --no-config1 --no-fp32-w48 --no-two-lanes --no-prepath --no-cpu-baseline > "$out/$tag.json" 2> "$out/$tag.err" < /dev/null
  fi
  python tools/abline.py "$tag" "$out/$tag.json" < /dev/null
done
