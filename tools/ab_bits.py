"""bitwise A/B of library variants: python tools/ab_bits.py  -> one line "tag sha256(heat-maps)" per call; run once per HRN_LIB_TAG
(tools/mkvariant.sh builds libhrnet_mi355_<tag>.so) and compare the hashes.  W48 384x288, 24 crops in micro-batches of 24 and 5."""
import hashlib, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("simple-hrnet_amd")
c, h, w = 48, 384, 288
x = torch.from_numpy(pkg.synth_crops(24, h, w, seed=41)).cuda()
hs = hashlib.sha256()
for mb in (24, 5):
    net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=mb, device=0).load_state_dict(pkg.synth_state_dict(c, 17, 0))
    hs.update(net(x).cpu().numpy().tobytes())
    net.close()
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn((256, 3, h, w), generator=g, device="cuda")
net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=256, device=0).load_state_dict(pkg.synth_state_dict(c, 17, 0))
hs.update(net(x).cpu().numpy().tobytes())
print(os.environ.get("HRN_LIB_TAG", "default"), hs.hexdigest()[:24])
