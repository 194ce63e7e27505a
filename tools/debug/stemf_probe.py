"""Debug aid for csrc/stem_fused.hip: conv2 := a one-tap channel-identity convolution, so that its output (tap "conv2") shows
conv1's output at the pixel that tap reads -- localises a difference between the fused kernel and the two launches to the
conv1 half (values / slab slots) or the conv2 half (K order, addresses)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from conftest import load_pkg, state_dict_np  # noqa: E402

pkg = load_pkg()
c, h, w, n = 48, 64, 64, 1
if len(sys.argv) > 3:
    h, w, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.from_numpy(pkg.synth_crops(n, h, w, seed=41)).cuda()


def run(sd, fused, tap="conv2"):
    if fused:
        os.environ.pop("HRN_DISABLE_STEM_FUSE", None)
    else:
        os.environ["HRN_DISABLE_STEM_FUSE"] = "1"
    net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=n, device=0).load_state_dict(sd)
    out = net.forward_tap(x, tap).cpu().numpy()
    net.close()
    return out


base = dict(state_dict_np(c))
stem = run(base, False, "stem")
print("conv1 output", stem.shape, "nonzero share %.2f" % (stem != 0).mean())
for dh in range(3):
    for dw in range(3):
        sd = dict(base)
        wt = np.zeros((64, 64, 3, 3), np.float32)
        wt[np.arange(64), np.arange(64), dh, dw] = 1.0
        sd["conv2.weight"] = wt
        sd["bn2.weight"] = np.ones(64, np.float32)
        sd["bn2.bias"] = np.zeros(64, np.float32)
        sd["bn2.running_mean"] = np.zeros(64, np.float32)
        sd["bn2.running_var"] = np.ones(64, np.float32) - 1e-5
        on, off = run(sd, True), run(sd, False)
        bad = on != off
        print("tap (%d, %d): mismatched %d / %d" % (dh, dw, bad.sum(), bad.size), end="")
        if bad.any():
            nn, cc, hh, ww = np.nonzero(bad)
            print("  channels %s rows %s cols %s" % (sorted(set(cc))[:20], sorted(set(hh))[:20], sorted(set(ww))[:20]))
            for k in range(min(6, len(nn))):
                i = (nn[k], cc[k], hh[k], ww[k])
                v = on[i]
                # where in conv1's output does the fused kernel's value come from?
                hits = np.argwhere(stem[nn[k]] == v) if v != 0 else []
                print("    at n=%d c=%d h=%d w=%d: fused %.5f two-launch %.5f; conv1 has the fused value at (c,h,w) %s" % (*i, v, off[i], [tuple(t) for t in hits[:4]]))
        else:
            print()
