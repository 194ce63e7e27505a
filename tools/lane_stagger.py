"""Experiment (round 5): two / three / four engines ("lanes") on ONE GPU, each on its own stream with a share of the 256-crop batch,
with lane k's start delayed on the GPU by k * d microseconds -- does putting the lanes out of phase (one in its HBM-bound fuse /
layer1 kernels while the other runs BasicBlock launches) buy more than the tail filling of lanes that start together?
usage: python tools/lane_stagger.py [steps]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("simple-hrnet_amd")
native = importlib.import_module("simple-hrnet_amd.native")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = 256
net = pkg.NativeHRNet(48, 17, (384, 288), "bf16", max_batch=N, device=0).load_state_dict(pkg.synth_state_dict(48, 17, 0))
g = torch.Generator(device="cuda").manual_seed(2)
x = torch.randn((N, 3, 384, 288), generator=g, device="cuda")
boxes = torch.tensor([[0, 0, 288, 384]] * N, dtype=torch.int32, device="cuda")
ticks_per_us = None


def timed(fn):
    for _ in range(2): r = fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps): r = fn()
    torch.cuda.synchronize()
    return N * steps / (time.perf_counter() - t), r


v0, p0 = timed(lambda: net.predict_crops(x, boxes))
print("one engine            %8.1f crops/s" % v0, flush=True)
# calibrate torch.cuda._sleep (cycles of the device's timer)
t = time.perf_counter(); torch.cuda._sleep(200_000_000); torch.cuda.synchronize(); cyc_per_us = 200_000_000 / ((time.perf_counter() - t) * 1e6)


class Staggered(native.MultiDeviceHRNet):
    delay_us = 0

    def _run(self, n, work):
        def work2(k, nt, lo, hi):
            if k and self.delay_us:
                torch.cuda._sleep(int(k * self.delay_us * cyc_per_us))
            return work(k, nt, lo, hi)
        return super()._run(n, work2)


for lanes in (2, 3, 4):
    eng = Staggered([0] * lanes, 48, 17, (384, 288), "bf16", max_batch=-(-N // lanes)).adopt_from(net)
    for d in ((0, 150, 300, 500, 800, 1200) if lanes == 2 else (0, 300)):
        eng.delay_us = d
        v, p = timed(lambda: eng.predict_crops(x, boxes))
        print("%d lanes, stagger %4d us %8.1f crops/s  x%.4f  same joints %s" % (lanes, d, v, v / v0, bool(torch.equal(p, p0))), flush=True)
    eng.close()
v1, _ = timed(lambda: net.predict_crops(x, boxes))
print("one engine again      %8.1f crops/s" % v1)
