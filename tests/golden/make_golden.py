#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE ITSELF.

Run in the build container only (needs /root/reference; the GPU box has neither
the reference nor this need -- it consumes the committed *.npz files):

    python tests/golden/make_golden.py

The reference has no tests, weights or golden vectors of its own,
so the pin for our oracle and for the HIP path is "outputs of the unmodified
reference code, executed here on CPU fp32, on seeded synthetic weights/inputs":

  * ``models_.hrnet.HRNet`` (hrnet.py:74-189) is imported unmodified and loaded with
    ``synth_state_dict`` -> heat-map fixtures.
  * ``SimpleHRNet.predict()`` (SimpleHRNet.py:174-496) is imported unmodified.  Its
    third-party imports that are absent from this image (cv2, torchvision, the
    un-vendored YOLOv3 submodule) are replaced by the small stand-ins below, which
    only touch the pre-path (colour flip, identity-size resize, PIL resize,
    ToTensor/Normalize) and the detector -- NOT the hot path under test (model call
    + decode, SimpleHRNet.py:281-308 / 416-443).  The crops the reference hands to
    ``self.model`` are captured and stored, so the parity tests feed identical bits.

Weights are not stored: they are regenerated from (c, seed) by
``simple-hrnet_amd/synth.py`` wherever the fixtures are used.
"""
from __future__ import annotations

import importlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

synth = importlib.import_module("simple-hrnet_amd.synth")


# ----------------------------------------------------------------------------- stand-ins (pre-path only)
def install_stubs():
    from PIL import Image

    cv2 = types.ModuleType("cv2")
    cv2.INTER_CUBIC = 2
    cv2.COLOR_BGR2RGB = 4

    def resize(img, dsize, interpolation=None):
        assert (img.shape[1], img.shape[0]) == tuple(dsize), "stub cv2.resize only supports identity size"
        return img

    def cvt_color(img, code):
        assert code == cv2.COLOR_BGR2RGB
        return np.ascontiguousarray(img[..., ::-1])

    cv2.resize = resize
    cv2.cvtColor = cvt_color
    sys.modules["cv2"] = cv2

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class ToPILImage:
        def __call__(self, a):
            return Image.fromarray(np.ascontiguousarray(a))

    class Resize:
        def __init__(self, size):
            self.size = size  # (h, w)

        def __call__(self, im):
            return im.resize((self.size[1], self.size[0]), Image.BILINEAR)

    class ToTensor:
        def __call__(self, a):
            a = np.asarray(a)
            return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).to(torch.float32).div(255)

    class Normalize:
        def __init__(self, mean, std):
            self.mean = torch.tensor(mean, dtype=torch.float32)[:, None, None]
            self.std = torch.tensor(std, dtype=torch.float32)[:, None, None]

        def __call__(self, t):
            return t.sub(self.mean).div(self.std)

    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    inner = types.ModuleType("torchvision.transforms.transforms")
    for m in (tvt, inner):
        m.Compose, m.ToPILImage, m.Resize, m.ToTensor, m.Normalize = Compose, ToPILImage, Resize, ToTensor, Normalize
    tvt.transforms = inner
    tv.transforms = tvt
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt
    sys.modules["torchvision.transforms.transforms"] = inner

    # detector stand-in: models_/detectors/YOLOv3.py:79-141 returns, per image, a (P,7) tensor
    # [x1,y1,x2,y2,conf,cls_conf,cls_pred] or None.
    det = types.ModuleType("models_.detectors.YOLOv3")

    class YOLOv3:
        table = {}

        def __init__(self, **kw):
            pass

        def predict_single(self, image, color_mode="BGR"):
            return self.predict(np.expand_dims(image, 0))[0]

        def predict(self, images, color_mode="BGR"):
            return [YOLOv3.table.get(i) for i in range(len(images))]

    det.YOLOv3 = YOLOv3
    sys.modules["models_.detectors.YOLOv3"] = det
    return YOLOv3


class Capture(torch.nn.Module):
    """wraps self.model to record what the reference feeds the hot path"""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner
        self.inputs = []

    def forward(self, x):
        self.inputs.append(x.detach().clone())
        return self.inner(x)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def ref_model(c, seed):
    from models_.hrnet import HRNet

    m = HRNet(c, 17).eval()
    m.load_state_dict(synth.to_torch_state_dict(synth.synth_state_dict(c, 17, seed)))
    return m


def ref_decode(out, boxes, h, w):
    """verbatim behaviour of SimpleHRNet.py:297-308, obtained by running predict() below;
    for the bare-HRNet fixtures we evaluate the same expressions here."""
    pts = np.empty((out.shape[0], out.shape[1], 3), dtype=np.float32)
    for i, human in enumerate(out):
        for j, joint in enumerate(human):
            pt = np.unravel_index(np.argmax(joint), (h, w))
            pts[i, j, 0] = pt[0] * 1. / h * (boxes[i][3] - boxes[i][1]) + boxes[i][1]
            pts[i, j, 1] = pt[1] * 1. / w * (boxes[i][2] - boxes[i][0]) + boxes[i][0]
            pts[i, j, 2] = joint[pt]
    return pts


def heatmap_case(name, c, n, h, w, seed=0):
    m = ref_model(c, seed)
    x = synth.synth_crops(n, h, w)
    with torch.no_grad():
        y = m(torch.from_numpy(x)).numpy()
    boxes = synth.synth_boxes(n)
    save(name, c=c, n=n, h=h, w=w, weight_seed=seed, heatmaps=y, boxes=boxes,
         pts=ref_decode(y, boxes, h // 4, w // 4))


def predict_cases(yolo):
    import SimpleHRNet as S

    tmp = tempfile.mkdtemp()

    def make(c, res, multiperson, seed=0, max_batch_size=32):
        ck = os.path.join(tmp, "w%d_%d.pth" % (c, seed))
        torch.save(synth.to_torch_state_dict(synth.synth_state_dict(c, 17, seed)), ck)
        s = S.SimpleHRNet(c, 17, ck, resolution=res, multiperson=multiperson, return_heatmaps=True,
                          return_bounding_boxes=True, max_batch_size=max_batch_size, device=torch.device("cpu"))
        s.model = Capture(s.model)
        return s

    rng = np.random.default_rng(0)

    # --- config 1 (BASELINE.json configs[0]): W32 256x192, ONE image with 3 people, device='cpu'
    frame = rng.integers(0, 256, (720, 1280, 3), dtype=np.uint8)
    yolo.table = {0: torch.tensor([[100.2, 50.7, 400.4, 650.1, .9, .9, 0.],      # tall box -> x padding
                                   [600.0, 200.0, 1100.0, 500.0, .8, .9, 0.],    # wide box -> y padding
                                   [-20.0, 300.0, 250.0, 700.0, .7, .9, 0.]])}   # touches the frame edge
    # (third box starts at x=-20 -> numpy negative-index slicing in the reference, kept as is)
    yolo.table[0][2, 0] = 5.0
    s = make(32, (256, 192), True)
    hm, boxes, pts = s.predict(frame)
    crops = torch.cat(s.model.inputs, 0).numpy()
    save("cfg1_w32_256x192_predict_multi", c=32, h=256, w=192, weight_seed=0, crops=crops, boxes=boxes,
         heatmaps=hm, pts=pts)

    # --- single-person path (float32 boxes, SimpleHRNet.py:213-225), frame == resolution
    frame = rng.integers(0, 256, (128, 96, 3), dtype=np.uint8)
    s = make(32, (128, 96), False)
    hm, boxes, pts = s.predict(frame)
    save("w32_128x96_predict_single", c=32, h=128, w=96, weight_seed=0, crops=s.model.inputs[0].numpy(),
         boxes=boxes, heatmaps=hm, pts=pts)

    # --- batch path, single-person, n=5 with max_batch_size=2 -> chunk loop (SimpleHRNet.py:423-429)
    frames = rng.integers(0, 256, (5, 128, 96, 3), dtype=np.uint8)
    s = make(48, (128, 96), False, max_batch_size=2)
    hm, boxes, pts = s.predict(frames)
    assert len(s.model.inputs) == 3
    save("w48_128x96_predict_batch5", c=48, h=128, w=96, weight_seed=0,
         crops=torch.cat(s.model.inputs, 0).numpy(), boxes=boxes, heatmaps=hm, pts=pts)

    # --- batch path, multi-person: image 0 has 2 people, image 1 none, image 2 one
    frames = rng.integers(0, 256, (3, 480, 640, 3), dtype=np.uint8)
    yolo.table = {0: torch.tensor([[50., 40., 200., 400., .9, .9, 0.], [300., 100., 620., 300., .9, .9, 0.]]),
                  1: None,
                  2: torch.tensor([[10., 10., 630., 470., .9, .9, 0.]])}
    s = make(32, (128, 96), True)
    hm, boxes, pts = s.predict(frames)
    counts = np.array([len(p) for p in pts], dtype=np.int32)
    save("w32_128x96_predict_batch_multi", c=32, h=128, w=96, weight_seed=0,
         crops=torch.cat(s.model.inputs, 0).numpy(), counts=counts,
         boxes=np.concatenate([np.asarray(b, dtype=np.int32).reshape(-1, 4) for b in boxes], 0),
         heatmaps=np.concatenate(hm, 0), pts=np.concatenate(pts, 0))


COCO_FLIP_PAIRS = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]  # datasets/COCO.py:113


def flip_tta_case(name, c, n, h, w, seed=0):
    """Flip test-time augmentation + evaluation decode, by the reference's own functions (testing/Test.py:132-140):
    ``flip_tensor`` / ``flip_back`` / ``get_final_preds`` of misc/utils.py, imported unmodified.  ``transform_preds``
    (the inverse affine, cv2.getAffineTransform) is replaced by the identity: our scope ends at heat-map coordinates."""
    if "munkres" not in sys.modules:
        sys.modules["munkres"] = types.ModuleType("munkres")
    import misc.utils as U

    m = ref_model(c, seed)
    x = torch.from_numpy(synth.synth_crops(n, h, w, seed=13))
    with torch.no_grad():
        out = m(x)
        out_f = U.flip_back(m(U.flip_tensor(x, dim=-1)), COCO_FLIP_PAIRS)
        avg = (out + out_f) * 0.5
    U.transform_preds = lambda coords, center, scale, pixel_std, output_size: coords
    none = [None] * n
    preds, maxvals = U.get_final_preds(True, avg.clone(), none, none, 200)
    preds_raw, _ = U.get_final_preds(False, avg.clone(), none, none, 200)
    save(name, c=c, n=n, h=h, w=w, weight_seed=seed, crops=x.numpy(), heatmaps=avg.numpy(), preds=preds.numpy(),
         preds_nopost=preds_raw.numpy(), maxvals=maxvals.numpy(), flip_pairs=np.asarray(COCO_FLIP_PAIRS, np.int32))


def poseresnet_case(name, size, n, h, w, seed=0):
    """models_/poseresnet.py PoseResNet, imported unmodified, on seeded synthetic weights."""
    from models_.poseresnet import PoseResNet

    m = PoseResNet(size, 17).eval()
    m.load_state_dict(synth.to_torch_state_dict(synth.synth_state_dict(size, 17, seed, model="PoseResNet")))
    x = synth.synth_crops(n, h, w, seed=17)
    with torch.no_grad():
        y = m(torch.from_numpy(x)).numpy()
    boxes = synth.synth_boxes(n)
    save(name, c=size, n=n, h=h, w=w, weight_seed=seed, crops=x, heatmaps=y, boxes=boxes,
         pts=ref_decode(y, boxes, h // 4, w // 4))


TAP_SAMPLES = 256


def tap_sample_index(size):
    """the elements of a flattened tap the fixture stores (tests/test_oracle.py recomputes the same index)"""
    return np.unique(np.linspace(0, size - 1, min(size, TAP_SAMPLES)).astype(np.int64))


def taps_case(name, c, n, h, w, seed=0):
    """Intermediate tensors of the reference's HRNet.forward (models_/hrnet.py:157-189), read with forward hooks on the
    unmodified modules and stored under the engine's tap names (include/hrnet_mi355.h: hrn_forward_tap) -- the pin for
    the NAMES and the semantics of the taps ("layer1.0.conv3" is the Bottleneck's output after the residual and the ReLU,
    "stage3.2.fuse.1" the second output of the third stage-3 module, a fuse-up 1x1 conv is read BEFORE its Upsample ...).
    Per tap: 256 evenly spaced elements + the fp64 sum and absolute sum of the whole tensor."""
    m = ref_model(c, seed)
    got = {}

    def keep(tap, relu=False):
        def hook(mod, inp, out):
            t = out.detach().clone()     # (the reference's ReLUs are in-place: copy before they run)
            got[tap] = torch.relu(t) if relu else t
        return hook

    def keep_list(prefix):
        def hook(mod, inp, out):
            for i, t in enumerate(out):
                got["%s.fuse.%d" % (prefix, i)] = t.detach().clone()
        return hook

    mods = dict(m.named_modules())
    mods["bn1"].register_forward_hook(keep("stem", relu=True))
    mods["bn2"].register_forward_hook(keep("conv2", relu=True))
    for path, mod in mods.items():
        cls = type(mod).__name__
        if cls == "Bottleneck":
            mods[path + ".bn1"].register_forward_hook(keep(path + ".conv1", relu=True))
            mods[path + ".bn2"].register_forward_hook(keep(path + ".conv2", relu=True))
            mod.register_forward_hook(keep(path + ".conv3"))
            if mod.downsample is not None:
                mods[path + ".downsample"].register_forward_hook(keep(path + ".downsample.0"))
        elif cls == "BasicBlock":
            mods[path + ".bn1"].register_forward_hook(keep(path + ".conv1", relu=True))
            mod.register_forward_hook(keep(path + ".conv2"))
        elif cls == "StageModule":
            mod.register_forward_hook(keep_list(path))
            nb = len(mod.branches)
            for i in range(len(mod.fuse_layers)):
                for j in range(nb):
                    q = "%s.fuse_layers.%d.%d" % (path, i, j)
                    if i < j:    # Sequential(conv1x1, bn, Upsample): the engine stores the BN output at low resolution
                        mods[q + ".1"].register_forward_hook(keep(q + ".0"))
                    elif i > j:  # Sequential of Sequential(conv3x3 s2, bn[, relu])
                        for k in range(i - j):
                            mods["%s.%d" % (q, k)].register_forward_hook(keep("%s.%d.0" % (q, k)))
    mods["transition1.0"].register_forward_hook(keep("transition1.0.0"))
    mods["transition1.1"].register_forward_hook(keep("transition1.1.0.0"))
    mods["transition2.2"].register_forward_hook(keep("transition2.2.0.0"))
    mods["transition3.3"].register_forward_hook(keep("transition3.3.0.0"))
    x = torch.from_numpy(synth.synth_crops(n, h, w, seed=21))
    with torch.no_grad():
        y = m(x)
    names = sorted(got)
    samples, sums, shapes = [], [], []
    for t in names:
        a = got[t].numpy().astype(np.float32).ravel()
        samples.append(a[tap_sample_index(a.size)])
        sums.append([a.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum()])
        shapes.append(list(got[t].shape))
    save(name, c=c, n=n, h=h, w=w, weight_seed=seed, crop_seed=21, names=np.asarray(names), heatmaps=y.numpy(),
         offsets=np.cumsum([0] + [len(s_) for s_ in samples]).astype(np.int64), samples=np.concatenate(samples),
         sums=np.asarray(sums, np.float64), shapes=np.asarray(shapes, np.int64))


def nms_boxes(n, seed, frame=(480, 640), crowd=True):
    """random detections: a few clusters of heavily overlapping boxes (what NMS is for) + scattered ones; distinct scores"""
    rng = np.random.default_rng(seed)
    centres = rng.uniform([40, 40], [frame[1] - 40, frame[0] - 40], size=(max(1, n // 8), 2))
    c = centres[rng.integers(0, len(centres), n)] + rng.normal(0, 12 if crowd else 80, (n, 2))
    wh = rng.uniform(20, 160, (n, 2))
    d = np.concatenate([c - wh / 2, c + wh / 2, rng.permutation(n)[:, None] / n + 0.001], 1)
    return d.astype(np.float32)


def nms_case(name):
    """misc/nms/nms.py:35-72 `nms`, imported unmodified (its compiled siblings cpu_nms / gpu_nms are not built here and
    are only imported at module top, so empty stand-in modules satisfy the import)."""
    for mod in ("cpu_nms", "gpu_nms"):
        m = types.ModuleType(mod)
        setattr(m, mod, None)
        sys.modules[mod] = m
    sys.path.insert(0, os.path.join(REF, "misc", "nms"))
    import nms as ref_nms

    arrays = {}
    for i, (n, thr, seed) in enumerate([(1, 0.5, 0), (7, 0.5, 1), (64, 0.3, 2), (65, 0.7, 3), (300, 0.5, 4), (1000, 0.45, 5)]):
        d = nms_boxes(n, seed)
        arrays["dets%d" % i] = d
        arrays["thresh%d" % i] = np.float32(thr)
        arrays["keep%d" % i] = np.asarray(ref_nms.nms(d, thr), dtype=np.int32)
    arrays["ncases"] = np.int32(6)
    save(name, **arrays)


def main():
    torch.set_num_threads(os.cpu_count())
    yolo = install_stubs()
    if len(sys.argv) > 1 and sys.argv[1] == "nms":
        nms_case("nms_cases")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "poseresnet":
        poseresnet_case("poseresnet50_128x96_n2", 50, 2, 128, 96, seed=3)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "taps":
        taps_case("w32_64x64_taps_n1", 32, 1, 64, 64)
        taps_case("w48_64x64_taps_n1", 48, 1, 64, 64, seed=1)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "fliptta":   # only the flip-TTA fixture
        flip_tta_case("w32_128x96_fliptta_n3", 32, 3, 128, 96, seed=2)
        return
    heatmap_case("w32_64x64_n2", 32, 2, 64, 64)
    heatmap_case("w48_64x64_n2", 48, 2, 64, 64)
    heatmap_case("w32_256x192_n2", 32, 2, 256, 192, seed=1)
    heatmap_case("w48_384x288_n1", 48, 1, 384, 288)
    predict_cases(yolo)
    flip_tta_case("w32_128x96_fliptta_n3", 32, 3, 128, 96, seed=2)
    poseresnet_case("poseresnet50_128x96_n2", 50, 2, 128, 96, seed=3)
    nms_case("nms_cases")
    taps_case("w32_64x64_taps_n1", 32, 1, 64, 64)
    taps_case("w48_64x64_taps_n1", 48, 1, 64, 64, seed=1)


if __name__ == "__main__":
    main()
