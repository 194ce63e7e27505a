"""layer1's launches of one 256-crop pass (HIP events via hrn_profile_pass), summed per Bottleneck and for the whole layer.
usage: python tools/layer1_time.py [mb]"""
import collections, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("simple-hrnet_amd")
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
net = pkg.NativeHRNet(48, 17, (384, 288), "bf16", max_batch=mb, device=0).load_state_dict(pkg.synth_state_dict(48, 17, 0))
x = torch.randn((mb, 3, 384, 288), device="cuda")
infos = net.conv_infos()
acc = None
for r in range(6):
    conv_ms, other = net.profile_pass(x)
    if r == 0: continue
    acc = conv_ms if acc is None else [p + q for p, q in zip(acc, conv_ms)]
conv_ms = [v / 5 for v in acc]
per = collections.OrderedDict()
for i, ms in zip(infos, conv_ms):
    nm = i.name.decode()
    if nm.startswith("layer1."):
        per.setdefault(nm.split(".")[1], []).append((nm, ms))
tot = 0.0
for b, items in per.items():
    s = sum(ms for _, ms in items); tot += s
    print("layer1.%s: %.3f ms  (%s)" % (b, s, ", ".join("%s %.3f" % (nm.split(".", 2)[2], ms) for nm, ms in items)))
print("layer1 total: %.3f ms; all convs %.3f ms; switches [%s]" % (tot, sum(conv_ms), net.switches()))
