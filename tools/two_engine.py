"""Does interleaving several engines on ONE GPU (their launches on separate streams fill each other's tails) beat one engine?
usage: python tools/two_engine.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("simple-hrnet_amd")
native = importlib.import_module("simple-hrnet_amd.native")
sd = pkg.synth_state_dict(48, 17, 0)
g = torch.Generator(device="cuda").manual_seed(1)


def bench(make, total, reps=6):
    net = make()
    crops = torch.randn((total, 3, 384, 288), generator=g, device="cuda")
    boxes = pkg.synth_boxes(total)
    for _ in range(2):
        net.predict_crops(crops, boxes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        pts = net.predict_crops(crops, boxes)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    net.close()
    return total / dt, pts


one, ref = bench(lambda: pkg.NativeHRNet(48, 17, (384, 288), "bf16", max_batch=256, device=0).load_state_dict(sd), 256)
print("one engine, 256 crops, micro-batch 256:        %8.1f crops/s" % one)
for k, mb, total in ((2, 128, 256), (2, 256, 512), (3, 128, 384), (4, 128, 512)):
    v, pts = bench(lambda: native.MultiDeviceHRNet([0] * k, 48, 17, (384, 288), "bf16", max_batch=mb).load_state_dict(sd), total)
    same = bool(torch.equal(pts[:256].cpu(), ref.cpu())) if total >= 256 else None
    print("%d engines on one GPU, %d crops, micro-batch %d: %8.1f crops/s  (same joints as one engine: %s)" % (k, total, mb, v, same))
v, _ = bench(lambda: pkg.NativeHRNet(48, 17, (384, 288), "bf16", max_batch=256, device=0).load_state_dict(sd), 512)
print("one engine, 512 crops, micro-batch 256:        %8.1f crops/s" % v)
