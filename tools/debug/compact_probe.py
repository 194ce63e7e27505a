"""compact on / off: first mismatching compact convolution (by tap) for a given geometry / batch / max_batch"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from conftest import load_pkg, state_dict_np  # noqa: E402

pkg = load_pkg()
h, w, n, mb = (int(a) for a in sys.argv[1:5])
x = torch.from_numpy(pkg.synth_crops(n, h, w, seed=61)).cuda()
res = {}
for tag in ("on", "off"):
    if tag == "off":
        os.environ["HRN_DISABLE_COMPACT"] = "1"
    else:
        os.environ.pop("HRN_DISABLE_COMPACT", None)
    net = pkg.NativeHRNet(48, 17, (h, w), "bf16", max_batch=mb, device=0).load_state_dict(state_dict_np(48))
    infos = net.conv_infos()
    if tag == "on":
        names = [i.name.decode() for k, i in enumerate(infos) if net.conv_compact(k)]
        grids = {i.name.decode(): (i.cin, i.out_h, i.out_w) for i in infos}
    res[tag] = {t: net.forward_tap(x, t).cpu().numpy() for t in names[:12]}
    res[tag + "_hm"] = net(x).cpu().numpy()
    net.close()
print("geometry %dx%d n=%d max_batch=%d: heat-maps equal: %s" % (h, w, n, mb, np.array_equal(res["on_hm"], res["off_hm"])))
for t in names[:12]:
    a, b = res["on"][t], res["off"][t]
    bad = a != b
    if bad.any():
        nn, cc, hh, ww = np.nonzero(bad)
        print("  %s %s: %d / %d differ; crops %s..%s, channels %s, rows %s, cols %s" % (t, grids[t], bad.sum(), bad.size, nn.min(), nn.max(), sorted(set(cc))[:6], sorted(set(hh))[:8], sorted(set(ww))[:8]))
        print("     bad crops (first 20):", sorted(set(nn))[:20])
        break
    else:
        print("  %s %s: equal" % (t, grids[t]))
