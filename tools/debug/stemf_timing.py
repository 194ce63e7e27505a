"""Per-phase shader-clock times of the fused stem kernel (needs the SF_TIMING variant: tools/mkvariant.sh sftime SF_TIMING,
run with HRN_LIB_TAG=sftime)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from conftest import load_pkg, state_dict_np  # noqa: E402

pkg = load_pkg()
c, h, w, n = 48, 384, 288, 256
net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=n, device=0).load_state_dict(state_dict_np(c))
x = torch.randn((n, 3, h, w), device="cuda")
lib = net._lib
lib.hrn_debug_stem_timing.restype = ctypes.c_int
lib.hrn_debug_stem_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(2):
    net(x)
out = (ctypes.c_uint64 * 16)()
assert lib.hrn_debug_stem_timing(out, 1) == 0
reps = 3
for _ in range(reps):
    net(x)
assert lib.hrn_debug_stem_timing(out, 1) == 0
names = ["phase A (conv1 -> slab)", "wait for the crop rows", "patch store + next loads", "barrier", "phase B (conv2)"]
for tag, o in (("wave 0", 0), ("wave 7", 8)):
    tiles = out[o + 5]
    tot = sum(out[o + i] for i in range(5))
    print("%s: %d tiles, %.0f clocks per tile" % (tag, tiles, tot / max(tiles, 1)))
    if out[o + 7]:
        print("   shader clock during the kernel: %.0f MHz (s_memtime / s_memrealtime x 100 MHz)" % (100.0 * out[o + 6] / out[o + 7]))
    for i, nm in enumerate(names):
        print("   %-26s %8.0f clocks / tile  %5.1f %%" % (nm, out[o + i] / max(tiles, 1), 100.0 * out[o + i] / max(tot, 1)))
