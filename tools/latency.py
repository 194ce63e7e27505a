"""Latency of small calls (BASELINE configs[0] shape: 3 person crops, W32 256x192)."""
import importlib, os, sys, time
NS = [int(v) for v in os.environ.get("LAT_NS", "1,3,16").split(",")]
CFGS = [tuple(int(x) for x in v.split("x")) for v in os.environ.get("LAT_CFGS", "32x256x192,48x384x288").split(",")]
DTS = os.environ.get("LAT_DTYPES", "fp32,bf16").split(",")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("simple-hrnet_amd")
for c, h, w in CFGS:
    for dtype in DTS:
        net = pkg.NativeHRNet(c, 17, (h, w), dtype, max_batch=max(NS), device=0).load_state_dict(pkg.synth_state_dict(c, 17, 0))
        for n in NS:
            x = torch.randn((n, 3, h, w), device="cuda")
            b = torch.from_numpy(pkg.synth_boxes(n)).cuda()
            for _ in range(5): net.predict_crops(x, b)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): net.predict_crops(x, b)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
            print("W%d %dx%d %s n=%d: %.2f ms per call (%.0f crops/s), %d launches" % (c, h, w, dtype, n, dt * 1e3, n / dt, net.launches_per_pass()))
        net.close()
