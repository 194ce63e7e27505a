"""Build a variant of the library with extra -D flags into libhrnet_mi355_<tag>.so and run bench.py against it.
usage: python tools/build_variant.py <tag> "<DEF1 DEF2=3>" [bench args...]"""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag, defs = sys.argv[1], sys.argv[2].split()
lib_mod = importlib.import_module("simple-hrnet_amd._lib")
for d in defs:
    lib_mod.HIPCC_FLAGS.append("-D" + d)
lib_mod.LIB_PATH = lib_mod.LIB_PATH.replace(".so", "_%s.so" % tag)
if not os.environ.get("NO_BUILD"):  # NO_BUILD=1: use the tagged .so as it is (e.g. one built from another commit)
    lib_mod.build(force=bool(os.environ.get("FORCE_BUILD")))   # the tag names the flag set: rebuilt only when sources are newer
sys.argv = ["bench.py"] + sys.argv[3:]
sys.path.insert(0, ROOT)
import runpy
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
