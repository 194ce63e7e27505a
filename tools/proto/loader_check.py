"""HRN_N96_LOADER on / off inside the whole net: heat-maps and joints must be bit-identical (same K order, same arithmetic)"""
import os, sys, importlib
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
pkg = importlib.import_module("simple-hrnet_amd")
from conftest import state_dict_np
outs = []
for n, h, w in ((96, 256, 192), (40, 384, 288)):
    crops = torch.from_numpy(pkg.synth_crops(n, h, w, seed=61)).cuda()
    boxes = pkg.synth_boxes(n, seed=62)
    res = []
    for mode in ("0", "6", "3"):
        os.environ["HRN_N96_LOADER"] = mode
        net = pkg.NativeHRNet(48, 17, (h, w), "bf16", max_batch=n, device=0).load_state_dict(state_dict_np(48))
        hm, pts = net.predict_crops(crops, boxes, return_heatmaps=True)
        res.append((hm.cpu().numpy(), pts.cpu().numpy()))
        net.close()
    for k in (1, 2):
        print((n, h, w), "mode", ("0", "6", "3")[k], "== off bit for bit:", np.array_equal(res[0][0], res[k][0]), np.array_equal(res[0][1], res[k][1]),
              "finite", np.isfinite(res[k][0]).all())
