"""The per-frame live loop (bench.py run_clip, mode per_frame) alone: wall time per frame; under
`rocprofv3 --kernel-trace` the trace gives the GPU-busy share (tools/r4_call12.sh sums it)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import bench  # noqa: E402
from conftest import load_pkg, state_dict_np  # noqa: E402

pkg = load_pkg()
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
net = pkg.NativeHRNet(48, 17, (384, 288), "bf16", max_batch=mb, device=0).load_state_dict(state_dict_np(48))
clip, dets = bench.make_clip()
clip_host = torch.from_numpy(clip).pin_memory()
for _ in range(2):
    bench.run_clip(net, clip_host, dets, "per_frame")
pts, el = bench.run_clip(net, clip_host, dets, "per_frame")
print("per_frame: %.3f ms per frame (%.1f fps), max_batch %d" % (el / clip.shape[0] * 1e3, clip.shape[0] / el, mb))
# GPU-only: the frames resident, crops pre-computed, hrn_forward alone
fdev = clip_host[0].to("cuda")
im, boxes, bdev = net.preprocess_frame(fdev, dets[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    net.predict_crops(im, bdev)
torch.cuda.synchronize()
print("forward alone on 8 resident crops: %.3f ms per call" % ((time.perf_counter() - t0) / 30 * 1e3))
t0 = time.perf_counter()
for _ in range(30):
    net.preprocess_frame(fdev, dets[0])
torch.cuda.synchronize()
print("pre-path alone: %.3f ms per call" % ((time.perf_counter() - t0) / 30 * 1e3))
ms, other = net.profile_pass(im)
print("profiled pass at n=8: convs %.3f ms, other %s, launches %d" % (sum(ms), {k: round(v, 3) for k, v in other.items()}, net.launches_per_pass()))
