"""Race hunt: the same call many times, every result compared bit for bit with the first (a counted wait that is one operation
too lenient, a tile read before its LDS-DMA landed or a stale block map shows up as a rare difference, not as a crash).
usage: python tools/stress_determinism.py [reps=150]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("simple-hrnet_amd")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
bad = 0
for c, h, w, dtype, n, mb in ((48, 384, 288, "bf16", 256, 256), (48, 384, 288, "bf16", 77, 256), (48, 384, 288, "bf16", 8, 8), (32, 256, 192, "bf16", 256, 256),
                              (32, 256, 192, "fp32", 64, 64), (48, 256, 192, "fp32", 40, 32)):
    net = pkg.NativeHRNet(c, 17, (h, w), dtype, max_batch=mb, device=0).load_state_dict(pkg.synth_state_dict(c, 17, 0))
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn((n, 3, h, w), generator=g, device="cuda")
    boxes = torch.from_numpy(pkg.synth_boxes(n, seed=1)).cuda()
    ref_hm, ref_pts = net.predict_crops(x, boxes, return_heatmaps=True)
    ref_hm, ref_pts = ref_hm.clone(), ref_pts.clone()
    k = max(4, reps * 64 // n if n < 256 else reps)
    diff = 0
    for it in range(k):
        hm, pts = net.predict_crops(x, boxes, return_heatmaps=True)
        if not (torch.equal(hm, ref_hm) and torch.equal(pts, ref_pts)):
            diff += 1
    torch.cuda.synchronize()
    print("W%d %dx%d %s n=%d mb=%d: %d repetitions, %d differ" % (c, h, w, dtype, n, mb, k, diff))
    bad += diff
    net.close()
sys.exit(1 if bad else 0)
