"""HIP-event time of every grouped BasicBlock launch of one pass (hrn_profile_pass splits a launch's time over its convolutions by
FLOPs: summed back per launch here), conv1-type launches (with the fused 48-channel blocks) against conv2-type ones."""
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
launch = collections.OrderedDict()
for i, ms in zip(infos, conv_ms):
    nm = i.name.decode()
    if ".branches." not in nm: continue
    f = nm.split(".")     # stageX.M.branches.B.K.convN
    key = (f[0], f[1], f[4], f[5])
    d = launch.setdefault(key, [0.0, 0.0])
    d[0] += ms; d[1] += i.flops * mb
tot = {}
for (st, m, k, cn), (ms, fl) in launch.items():
    t = tot.setdefault((st, cn), [0, 0.0, 0.0]); t[0] += 1; t[1] += ms; t[2] += fl
for (st, cn), (n, ms, fl) in tot.items():
    print("%s %s: %2d launches, %.1f us each, %.0f TFLOP/s algorithmic" % (st, cn, n, 1e3 * ms / n, fl / ms / 1e9))
