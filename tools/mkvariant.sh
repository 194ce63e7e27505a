#!/bin/bash
# build libhrnet_mi355_<tag>.so with extra -D flags (quiet; log in /tmp/build_<tag>.log): tools/mkvariant.sh <tag> "<DEF1 DEF2=3>"
tag=$1; defs=$2
python - "$tag" "$defs" > /tmp/build_$tag.log 2>&1 <<'PY'
import importlib, sys
sys.path.insert(0, '/root/repo')
tag, defs = sys.argv[1], sys.argv[2].split()
m = importlib.import_module("simple-hrnet_amd._lib")
for d in defs: m.HIPCC_FLAGS.append("-D" + d)
if tag != "default":
    m.LIB_PATH = m.LIB_PATH.replace(".so", "_%s.so" % tag)
m.build(force=True)
print("built", tag)
PY
tail -n 1 /tmp/build_$tag.log
