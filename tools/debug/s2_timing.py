"""Where a tile of the stride-2 slab kernel spends its time (needs the S2_TIMING variant: tools/mkvariant.sh s2time S2_TIMING,
run with HRN_LIB_TAG=s2time): waiting for the slab + barrier against everything else, per tile; block prologue."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from conftest import load_pkg, state_dict_np  # noqa: E402

pkg = load_pkg()
c, h, w, n = 48, 384, 288, 256
os.environ["HRN_DISABLE_STEM_FUSE"] = "1"      # (the fused stem is another kernel; keep this reading to the slab kernel's launches)
net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=n, device=0).load_state_dict(state_dict_np(c))
x = torch.randn((n, 3, h, w), device="cuda")
lib = net._lib
lib.hrn_debug_s2_timing.restype = ctypes.c_int
lib.hrn_debug_s2_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(2):
    net(x)
out = (ctypes.c_uint64 * 8)()
assert lib.hrn_debug_s2_timing(out, 1) == 0
reps = 3
for _ in range(reps):
    net(x)
assert lib.hrn_debug_s2_timing(out, 1) == 0
tiles, blocks = out[2], out[4]
print("%d blocks, %d tiles over %d passes (15 launches each)" % (blocks, tiles, reps))
print("per tile: waiting for the slab + barrier %.0f clocks, the rest %.0f clocks" % (out[0] / tiles, out[1] / tiles))
print("per block: prologue (weights -> registers, first slab issued) %.0f clocks, tiles %.1f, loop total %.0f clocks" % (
    out[3] / blocks, tiles / blocks, (out[0] + out[1]) / blocks))
