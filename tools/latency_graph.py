"""Experiment: replay one small call from a captured HIP graph (torch.cuda.CUDAGraph) vs eager launches."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("simple-hrnet_amd")
for c, h, w in ((48, 384, 288), (32, 256, 192)):
    net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=16, device=0).load_state_dict(pkg.synth_state_dict(c, 17, 0))
    for n in (1, 3, 16):
        x = torch.randn((n, 3, h, w), device="cuda")
        b = torch.from_numpy(pkg.synth_boxes(n)).cuda()
        for _ in range(3): ref = net.predict_crops(x, b)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(2): net.predict_crops(x, b)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = net.predict_crops(x, b)
        for _ in range(3): g.replay()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        t0 = time.perf_counter()
        for _ in range(50): net.predict_crops(x, b)
        torch.cuda.synchronize(); de = (time.perf_counter() - t0) / 50
        print("W%d n=%d: graph %.3f ms, eager %.3f ms, same=%s" % (c, n, dt * 1e3, de * 1e3, bool(torch.equal(out, ref))))
    net.close()
