"""Per-convolution times of one pass (hrn_profile_pass: HIP events around every launch, a grouped launch's time split over its
members by FLOPs) -- the convolutions OUTSIDE the graded BasicBlock kernel, ranked: where the 'other third' of the pass is.
usage: python tools/conv_table.py [c h w n]"""
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
from conftest import load_pkg, state_dict_np  # noqa: E402

pkg = load_pkg()
c, h, w, n = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (48, 384, 288, 256)
net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=n, device=0).load_state_dict(state_dict_np(c))
x = torch.randn((n, 3, h, w), device="cuda")
for _ in range(2):
    net(x)
acc = None
reps = 5
for _ in range(reps):
    ms, other = net.profile_pass(x)
    acc = ms if acc is None else [a + b for a, b in zip(acc, ms)]
ms = [a / reps for a in acc]
infos = net.conv_infos()
ALGO = {0: "generic", 1: "lds", 2: "fused-bb", 3: "n96", 4: "slab"}
groups = defaultdict(lambda: [0, 0.0, 0.0])
rows = []
for i, ci in enumerate(infos):
    if ci.name == b"conv2" and net.stem_fused():
        continue   # runs inside stem_fused_kernel (listed under "other": stem); its slot in the per-conv times is an empty launch
    fl = 2.0 * ci.cout * ci.cin * ci.ksize * ci.ksize * ci.out_h * ci.out_w * n
    key = (ALGO.get(ci.algo, str(ci.algo)), "%dx%d s%d" % (ci.ksize, ci.ksize, ci.stride), ci.cin, ci.cout, ci.out_h)
    g = groups[key]
    g[0] += 1
    g[1] += ms[i]
    g[2] += fl
    rows.append((ms[i], ci.name.decode(), key))
print("other:", {k: round(v, 3) for k, v in other.items()}, " sum of convs %.2f ms" % sum(ms))
print("%-9s %-7s %5s %5s %5s %4s %8s %8s %7s" % ("kernel", "conv", "cin", "cout", "out_h", "n", "ms", "GFLOP", "TF"))
for key, g in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    print("%-9s %-7s %5d %5d %5d %4d %8.3f %8.1f %7.0f" % (*key, g[0], g[1], g[2] / 1e9, g[2] / max(g[1], 1e-9) / 1e9))
