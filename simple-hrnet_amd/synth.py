"""Deterministic synthetic HRNet checkpoints and crops.

There is no network on the build or GPU boxes, so neither the official
``pose_hrnet_*.pth`` weights nor real images are available.  Everything that
needs weights (parity tests, ``bench.py``, ``__graft_entry__.smoke``) uses the
generator below.  It is *reference-independent*: it enumerates the 1754
``state_dict`` entries of the reference network by name and shape
(``/root/reference/models_/hrnet.py:75-155``, ``models_/modules.py:5-72``) and
fills them from a counter-based numpy RNG, so the very same checkpoint can be
rebuilt on the GPU box (where ``/root/reference`` does not exist) and loaded
into the reference ``HRNet`` in the build container
(``tests/golden/make_golden.py``).

BatchNorm statistics are randomised on purpose: with default-initialised BN the
fold (``W' = W * g / sqrt(v + eps)``) is numerically invisible, and a folding
bug would pass every test (SURVEY.md §7 "BN-fold visibility").
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np

BN_EPS = 1e-5  # nn.BatchNorm2d default, hrnet.py:80 / modules.py:11


def _bn(prefix: str, ch: int) -> List[Tuple[str, Tuple[int, ...], str]]:
    return [
        (prefix + ".weight", (ch,), "bn_gamma"),
        (prefix + ".bias", (ch,), "bn_beta"),
        (prefix + ".running_mean", (ch,), "bn_mean"),
        (prefix + ".running_var", (ch,), "bn_var"),
        (prefix + ".num_batches_tracked", (), "bn_count"),
    ]


def _conv(name: str, cout: int, cin: int, k: int) -> List[Tuple[str, Tuple[int, ...], str]]:
    return [(name + ".weight", (cout, cin, k, k), "conv")]


def hrnet_state_spec(c: int = 48, nof_joints: int = 17) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, kind) for every entry of ``HRNet(c, nof_joints).state_dict()``.

    Mirrors the module tree of hrnet.py:75-155 (stem, layer1, transition1-3,
    stage2-4, final_layer) and hrnet.py:7-53 (StageModule branches + fuse
    layers).  Verified against the reference in tests/test_synth.py.
    """
    s: List[Tuple[str, Tuple[int, ...], str]] = []
    # stem, hrnet.py:79-83
    s += _conv("conv1", 64, 3, 3) + _bn("bn1", 64)
    s += _conv("conv2", 64, 64, 3) + _bn("bn2", 64)
    # layer1 = 4 x Bottleneck(.., 64), hrnet.py:86-95, modules.py:8-16
    for b in range(4):
        cin = 64 if b == 0 else 256
        p = "layer1.%d" % b
        s += _conv(p + ".conv1", 64, cin, 1) + _bn(p + ".bn1", 64)
        s += _conv(p + ".conv2", 64, 64, 3) + _bn(p + ".bn2", 64)
        s += _conv(p + ".conv3", 256, 64, 1) + _bn(p + ".bn3", 256)
        if b == 0:
            s += _conv(p + ".downsample.0", 256, 64, 1) + _bn(p + ".downsample.1", 256)
    # transition1, hrnet.py:98-109
    s += _conv("transition1.0.0", c, 256, 3) + _bn("transition1.0.1", c)
    s += _conv("transition1.1.0.0", 2 * c, 256, 3) + _bn("transition1.1.0.1", 2 * c)

    def stage(name: str, nbranch: int, nout: int) -> None:
        # StageModule, hrnet.py:7-53
        for b in range(nbranch):
            w = c << b
            for k in range(4):
                p = "%s.branches.%d.%d" % (name, b, k)
                s.extend(_conv(p + ".conv1", w, w, 3) + _bn(p + ".bn1", w))
                s.extend(_conv(p + ".conv2", w, w, 3) + _bn(p + ".bn2", w))
        for i in range(nout):
            for j in range(nbranch):
                p = "%s.fuse_layers.%d.%d" % (name, i, j)
                if i < j:  # 1x1 conv + BN (+ nearest upsample), hrnet.py:30-35
                    s.extend(_conv(p + ".0", c << i, c << j, 1) + _bn(p + ".1", c << i))
                elif i > j:  # chain of 3x3 s2 convs, hrnet.py:36-51
                    for k in range(i - j - 1):
                        s.extend(_conv("%s.%d.0" % (p, k), c << j, c << j, 3))
                        s.extend(_bn("%s.%d.1" % (p, k), c << j))
                    k = i - j - 1
                    s.extend(_conv("%s.%d.0" % (p, k), c << i, c << j, 3))
                    s.extend(_bn("%s.%d.1" % (p, k), c << i))

    stage("stage2.0", 2, 2)  # hrnet.py:112-114
    s += _conv("transition2.2.0.0", 4 * c, 2 * c, 3) + _bn("transition2.2.0.1", 4 * c)
    for m in range(4):  # hrnet.py:128-133
        stage("stage3.%d" % m, 3, 3)
    s += _conv("transition3.3.0.0", 8 * c, 4 * c, 3) + _bn("transition3.3.0.1", 8 * c)
    stage("stage4.0", 4, 4)  # hrnet.py:148-152
    stage("stage4.1", 4, 4)
    stage("stage4.2", 4, 1)
    # final_layer: 1x1 conv WITH bias, hrnet.py:155
    s += [("final_layer.weight", (nof_joints, c, 1, 1), "conv"),
          ("final_layer.bias", (nof_joints,), "conv_bias")]
    return s


RESNET_SPEC = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}  # poseresnet.py:6-12 (Bottleneck sizes)


def poseresnet_state_spec(resnet_size: int = 50, nof_joints: int = 17) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, kind) for every entry of ``PoseResNet(resnet_size, nof_joints).state_dict()``
    (models_/poseresnet.py:16-122; Bottleneck of models_/modules.py:5-40).  Sizes 18 / 34 are not offered: the
    reference's ``BasicBlock`` builds ``conv2`` with ``inplanes`` input channels (modules.py:50) and fails at
    run time as soon as a layer changes width."""
    layers = RESNET_SPEC[resnet_size]
    s: List[Tuple[str, Tuple[int, ...], str]] = []
    s += _conv("conv1", 64, 3, 7) + _bn("bn1", 64)
    inplanes = 64
    for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), layers)):
        for b in range(blocks):
            p = "layer%d.%d" % (li + 1, b)
            s += _conv(p + ".conv1", planes, inplanes, 1) + _bn(p + ".bn1", planes)
            s += _conv(p + ".conv2", planes, planes, 3) + _bn(p + ".bn2", planes)
            s += _conv(p + ".conv3", planes * 4, planes, 1) + _bn(p + ".bn3", planes * 4)
            if b == 0:  # stride != 1 or inplanes != planes * 4 (poseresnet.py:53-59)
                s += _conv(p + ".downsample.0", planes * 4, inplanes, 1) + _bn(p + ".downsample.1", planes * 4)
            inplanes = planes * 4
    for i in range(3):  # ConvTranspose2d(in, 256, 4, stride 2, padding 1, bias=False) + BN + ReLU
        s += [("deconv_layers.%d.weight" % (3 * i), (inplanes, 256, 4, 4), "deconv")] + _bn("deconv_layers.%d" % (3 * i + 1), 256)
        inplanes = 256
    s += [("final_layer.weight", (nof_joints, 256, 1, 1), "conv"), ("final_layer.bias", (nof_joints,), "conv_bias")]
    return s


def _rng_for(seed: int, key: str) -> np.random.Generator:
    # one independent stream per tensor, keyed by name: order-independent
    return np.random.default_rng([seed, zlib.crc32(key.encode())])


def synth_state_dict(c: int = 48, nof_joints: int = 17, seed: int = 0, model: str = "HRNet") -> "OrderedDict[str, np.ndarray]":
    """Seeded checkpoint as numpy arrays (fp32; ``num_batches_tracked`` int64).  ``model="PoseResNet"``: ``c`` is
    the ResNet size (50 / 101 / 152).

    conv: U(-b, b), b = 1/sqrt(fan_in)  (the bound torch's default Conv2d init uses)
    BN:   gamma ~ U(0.5,1.5), beta ~ N(0,0.1), mean ~ N(0,0.1), var ~ U(0.5,1.5)
    """
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    spec = hrnet_state_spec(c, nof_joints) if model in ("HRNet", "hrnet") else poseresnet_state_spec(c, nof_joints)
    for key, shape, kind in spec:
        g = _rng_for(seed, key)
        if kind == "deconv":   # (cin, cout, 4, 4): every output sees cin x 2 x 2 taps
            b = 1.0 / np.sqrt(shape[0] * 4)
            a = g.uniform(-b, b, size=shape)
        elif kind == "conv":
            fan_in = shape[1] * shape[2] * shape[3]
            b = 1.0 / np.sqrt(fan_in)
            a = g.uniform(-b, b, size=shape)
        elif kind == "conv_bias":
            b = 1.0 / np.sqrt(c if model in ("HRNet", "hrnet") else 256)
            a = g.uniform(-b, b, size=shape)
        elif kind == "bn_gamma":
            a = g.uniform(0.5, 1.5, size=shape)
        elif kind in ("bn_beta", "bn_mean"):
            a = g.normal(0.0, 0.1, size=shape)
        elif kind == "bn_var":
            a = g.uniform(0.5, 1.5, size=shape)
        elif kind == "bn_count":
            out[key] = np.asarray(1, dtype=np.int64)
            continue
        else:  # pragma: no cover
            raise AssertionError(kind)
        out[key] = np.ascontiguousarray(a, dtype=np.float32)
    return out


def synth_crops(n: int, height: int, width: int, seed: int = 2) -> np.ndarray:
    """(n,3,H,W) fp32 crops in the post-normalisation domain (~N(0,1)), the
    tensor ``SimpleHRNet`` hands to ``self.model`` (SimpleHRNet.py:277-286)."""
    g = np.random.default_rng([seed, n, height, width])
    return g.standard_normal((n, 3, height, width), dtype=np.float32)


def synth_boxes(n: int, seed: int = 3, frame_hw: Tuple[int, int] = (1080, 1920)) -> np.ndarray:
    """(n,4) int32 ``[x1,y1,x2,y2]`` person boxes as ``_predict_single`` stores them
    (SimpleHRNet.py:230,278): may be negative / exceed the frame after padding."""
    g = np.random.default_rng([seed, n])
    h = g.integers(300, 900, size=n)
    w = (h * 3) // 4
    x1 = g.integers(-50, frame_hw[1] - 100, size=n)
    y1 = g.integers(-50, frame_hw[0] - 100, size=n)
    return np.stack([x1, y1, x1 + w, y1 + h], axis=1).astype(np.int32)


def to_torch_state_dict(sd: Dict[str, np.ndarray]):
    import torch

    return OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in sd.items())


# ----------------------------------------------------------------------------------------------- peaked checkpoint
# A random-init HRNet produces near-flat heat-maps (median top-1 / top-2 gap 1.7e-2 at sigma 5.7): their
# arg-max is decided by the last bits, so "same joint coordinates" cannot be asserted for the bf16 engine on them.  A
# trained network has ONE dominant blob per joint.  No trained weights exist offline, and a linear read-out of the random
# trunk's final features cannot localise anything (tried: least squares of a Gaussian target on the branch-0 features of
# 16 marked crops peaks at the marker for 1.5 % of the joints -- a deep random trunk scrambles position), so the peaked
# checkpoint is BUILT: the random checkpoint above with three "signal" channels laid through the identity paths of the
# graph (stem -> layer1's shortcut -> transition1.0 -> the residual connections of branch 0 -> the i == j terms of the
# fuse sums, hrnet.py:60-69), every weight that writes into those channels from elsewhere scaled by `leak` (NOT zeroed:
# the channels pick up the random network's activity and its bf16 rounding noise at every one of the ~40 stored tensors
# on the way), and a final layer that reads each joint from a mix of the three signals plus the usual random weights on
# the other channels.  Crops for it carry one positive blob per colour plane on weak noise; joint j then peaks where the
# strongest weighted blob sits, with a margin to the neighbouring cell of ~12 % of the peak (sigma = 2 cells).
PEAKED_SIGNALS = 3


def peaked_state_dict(c: int = 48, nof_joints: int = 17, seed: int = 0, leak: float = 0.05,
                      head_noise: float = 0.25) -> "OrderedDict[str, np.ndarray]":
    """The seeded random checkpoint with three signal channels (see above).  Pure numpy, reference independent."""
    sd = synth_state_dict(c, nof_joints, seed)
    S = PEAKED_SIGNALS

    def ident_bn(prefix, ch=S):          # y = x (up to the eps in the denominator)
        sd[prefix + ".weight"][:ch] = 1.0
        sd[prefix + ".bias"][:ch] = 0.0
        sd[prefix + ".running_mean"][:ch] = 0.0
        sd[prefix + ".running_var"][:ch] = 1.0

    def leak_into(conv, bn, ch=S):       # out channels 0..ch-1 of (conv, bn): only a leak of the random activity
        sd[conv + ".weight"][:ch] *= leak
        sd[bn + ".bias"][:ch] *= leak
        sd[bn + ".running_mean"][:ch] *= leak

    # stem: a 3x3 mean of colour plane s, twice (stride 2 each) -> 1/4 resolution
    for conv, bn, cin in (("conv1", "bn1", 3), ("conv2", "bn2", 64)):
        w = sd[conv + ".weight"]
        w[:S] *= leak
        for s in range(S):
            w[s, s] += 1.0 / 9.0
        ident_bn(bn)
    # layer1: the projection shortcut of block 0 carries the signals, the other blocks' identity shortcuts keep them
    w = sd["layer1.0.downsample.0.weight"]
    w[:S] *= leak
    for s in range(S):
        w[s, s, 0, 0] += 1.0
    ident_bn("layer1.0.downsample.1")
    for b in range(4):
        leak_into("layer1.%d.conv3" % b, "layer1.%d.bn3" % b)
    # transition1.0: centre tap
    w = sd["transition1.0.0.weight"]
    w[:S] *= leak
    for s in range(S):
        w[s, s, 1, 1] += 1.0
    ident_bn("transition1.0.1")
    # branch 0 of every stage module: the residual connection carries the signals; fuse terms into output 0 only leak
    for name, nbranch in (("stage2.0", 2), ("stage3.0", 3), ("stage3.1", 3), ("stage3.2", 3), ("stage3.3", 3),
                          ("stage4.0", 4), ("stage4.1", 4), ("stage4.2", 4)):
        for k in range(4):
            leak_into("%s.branches.0.%d.conv2" % (name, k), "%s.branches.0.%d.bn2" % (name, k))
        for j in range(1, nbranch):
            leak_into("%s.fuse_layers.0.%d.0" % (name, j), "%s.fuse_layers.0.%d.1" % (name, j))
    # head: joint j reads signal j % 3 with weight 1 and the others with U(0, 0.4); the random weights on the other
    # channels stay, scaled so that their field (sigma ~ 1.6 unscaled) cannot reach the blobs' height
    g = _rng_for(seed, "peaked.head")
    w = sd["final_layer.weight"]
    w *= head_noise
    mix = g.uniform(0.0, 0.4, size=(nof_joints, S)).astype(np.float32)
    for j in range(nof_joints):
        mix[j, j % S] = 1.0
    w[:, :S, 0, 0] = mix
    sd["final_layer.bias"] *= head_noise
    return sd


def peaked_crops(n: int, height: int, width: int, seed: int = 5, sigma_px: float = 6.0, noise: float = 0.2, on_cell: bool = True):
    """(n,3,H,W) fp32 crops for `peaked_state_dict`: per colour plane ONE positive Gaussian blob (sigma 6 input pixels = 1.5
    heat-map cells, amplitude U(20, 30), centre seeded, at least 16 px from the border and from the other planes' blobs)
    on N(0, 0.2) noise.  ``on_cell``: centres are multiples of 4 -- the input pixel heat-map cell (Y, X) is centred on
    (two 3x3 / stride-2 / pad-1 convolutions: 4Y, 4X) -- so the peak cell beats its neighbours by ~20 % of its height;
    off-cell centres can sit between two cells, whose values then tie to within the bf16 noise whatever the network
    (the reference decodes without sub-cell refinement, SimpleHRNet.py:297-308: one cell = 4 px of the crop).
    Returns (crops, centres (n,3,2) int: y, x in input pixels)."""
    g = np.random.default_rng([seed, n, height, width])
    x = (noise * g.standard_normal((n, 3, height, width))).astype(np.float32)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    centres = np.zeros((n, PEAKED_SIGNALS, 2), np.int64)
    for i in range(n):
        chosen = []
        for s in range(PEAKED_SIGNALS):
            for _ in range(1000):
                cy, cx = int(g.integers(16, height - 16)), int(g.integers(16, width - 16))
                if on_cell:
                    cy, cx = cy // 4 * 4, cx // 4 * 4
                if all(max(abs(cy - a), abs(cx - b)) >= 16 for a, b in chosen):
                    break
            chosen.append((cy, cx))
            amp = float(g.uniform(20.0, 30.0))
            x[i, s] += amp * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2.0 * sigma_px * sigma_px))
        centres[i] = chosen
    return x, centres
