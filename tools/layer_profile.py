"""Per-shape kernel time table of one internal pass (HIP events via hrn_profile_pass)."""
import argparse, collections, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--c", type=int, default=48); ap.add_argument("--height", type=int, default=384)
ap.add_argument("--width", type=int, default=288); ap.add_argument("--dtype", default="bf16")
ap.add_argument("--mb", type=int, default=64); ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--model-name", default="HRNet")
a = ap.parse_args()
pkg = importlib.import_module("simple-hrnet_amd")
net = pkg.NativeHRNet(a.c, 17, (a.height, a.width), a.dtype, max_batch=a.mb, device=0, model_name=a.model_name).load_state_dict(pkg.synth_state_dict(a.c, 17, 0, model=a.model_name))
x = torch.randn((a.mb, 3, a.height, a.width), device="cuda")
infos = net.conv_infos()
acc = None
for r in range(a.reps + 1):
    conv_ms, other = net.profile_pass(x)
    if r == 0: continue
    acc = conv_ms if acc is None else [p + q for p, q in zip(acc, conv_ms)]
conv_ms = [v / a.reps for v in acc]
groups = collections.OrderedDict()
for i, ms in zip(infos, conv_ms):
    key = (i.ksize, i.stride, i.cin, i.cout, i.out_h, i.out_w, i.algo)
    g = groups.setdefault(key, [0, 0.0, 0.0])
    g[0] += 1; g[1] += ms; g[2] += i.flops * a.mb
tot = sum(conv_ms) + sum(other.values())
print("pass of %d crops: %.3f ms total (convs %.3f, other %s)" % (a.mb, tot, sum(conv_ms), {k: round(v, 3) for k, v in other.items()}))
print("%-34s %4s %9s %7s %9s %8s" % ("k s cin->cout @HxW algo", "n", "ms", "%pass", "TFLOP/s", "us/launch"))
for key, (n, ms, fl) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    k, s, ci, co, h, w, al = key
    print("%-34s %4d %9.3f %6.1f%% %9.1f %8.1f" % ("%dx%d s%d %3d->%3d @%dx%d a%d" % (k, k, s, ci, co, h, w, al), n, ms, 100 * ms / tot, fl / ms / 1e9, 1e3 * ms / n))
