"""ORACLE (test infrastructure, NOT product code) -- torch-CPU restatement of the
reference hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product (``simple-hrnet_amd``) never does.

What it restates (citations into /root/reference):
  * ``HRNet.forward``            models_/hrnet.py:157-189
  * ``StageModule.forward``      models_/hrnet.py:55-71  (fuse order j = 0..B-1)
  * ``Bottleneck.forward``       models_/modules.py:20-40
  * ``BasicBlock.forward``       models_/modules.py:56-72
  * heat-map decode loop         SimpleHRNet.py:297-308 (dup. 432-443)
  * ``max_batch_size`` chunking  SimpleHRNet.py:284-296

The reference's arithmetic lives in PyTorch itself (nn.Conv2d / BatchNorm2d /
ReLU / Upsample -> ATen -> oneDNN on CPU), so this restatement walks the
``state_dict`` with the same ``torch.nn.functional`` primitives in the same
order; on CPU fp32 it reproduces the reference bit-for-bit (pinned by
tests/test_oracle.py against the real ``models_.hrnet.HRNet`` when
/root/reference is present, and by the committed fixtures in tests/golden/
everywhere else).  Parity status: the reference ships no tests or golden
vectors of its own -- the pin is "outputs of the reference
itself, run in the build container" (tests/golden/make_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _t(sd: Dict, k: str) -> torch.Tensor:
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def _conv(sd, name, x, stride=1):
    w = _t(sd, name + ".weight")
    k = w.shape[-1]
    b = sd.get(name + ".bias")
    b = None if b is None else _t(sd, name + ".bias")
    return F.conv2d(x, w, b, stride=stride, padding=k // 2)


def _bn(sd, name, x):
    # eval-mode BatchNorm2d: y = (x - mean) / sqrt(var + eps) * gamma + beta
    return F.batch_norm(x, _t(sd, name + ".running_mean"), _t(sd, name + ".running_var"),
                        _t(sd, name + ".weight"), _t(sd, name + ".bias"), False, 0.0, BN_EPS)


def bottleneck(sd, p, x, has_downsample):
    """modules.py:20-40."""
    out = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x)))
    out = F.relu(_bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out)))
    out = _bn(sd, p + ".bn3", _conv(sd, p + ".conv3", out))
    res = x
    if has_downsample:
        res = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x))
    return F.relu(out + res)


def basic_block(sd, p, x):
    """modules.py:56-72."""
    out = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x)))
    out = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out))
    return F.relu(out + x)


def stage_module(sd, p, xs: List[torch.Tensor], nout: int) -> List[torch.Tensor]:
    """hrnet.py:55-71."""
    nb = len(xs)
    ys = []
    for b in range(nb):
        y = xs[b]
        for k in range(4):
            y = basic_block(sd, "%s.branches.%d.%d" % (p, b, k), y)
        ys.append(y)
    fused = []
    for i in range(nout):
        acc = None
        for j in range(nb):
            q = "%s.fuse_layers.%d.%d" % (p, i, j)
            if i == j:
                t = ys[j]
            elif i < j:  # hrnet.py:30-35
                t = _bn(sd, q + ".1", _conv(sd, q + ".0", ys[j]))
                t = F.interpolate(t, scale_factor=float(2 ** (j - i)), mode="nearest")
            else:  # hrnet.py:36-51
                t = ys[j]
                for k in range(i - j - 1):
                    t = F.relu(_bn(sd, "%s.%d.1" % (q, k), _conv(sd, "%s.%d.0" % (q, k), t, stride=2)))
                k = i - j - 1
                t = _bn(sd, "%s.%d.1" % (q, k), _conv(sd, "%s.%d.0" % (q, k), t, stride=2))
            acc = t if j == 0 else acc + t  # hrnet.py:63-66: left-to-right accumulation
        fused.append(F.relu(acc))
    return fused


@torch.no_grad()
def hrnet_forward(sd: Dict, images) -> torch.Tensor:
    """hrnet.py:157-189.  images: (N,3,H,W) fp32 -> heat-maps (N,J,H/4,W/4)."""
    x = images if isinstance(images, torch.Tensor) else torch.from_numpy(np.asarray(images))
    dt = _t(sd, "conv1.weight").dtype
    x = x.to(dt)
    x = F.relu(_bn(sd, "bn1", _conv(sd, "conv1", x, stride=2)))
    x = F.relu(_bn(sd, "bn2", _conv(sd, "conv2", x, stride=2)))
    for b in range(4):
        x = bottleneck(sd, "layer1.%d" % b, x, b == 0)
    xs = [
        F.relu(_bn(sd, "transition1.0.1", _conv(sd, "transition1.0.0", x))),
        F.relu(_bn(sd, "transition1.1.0.1", _conv(sd, "transition1.1.0.0", x, stride=2))),
    ]
    xs = stage_module(sd, "stage2.0", xs, 2)
    xs = xs + [F.relu(_bn(sd, "transition2.2.0.1", _conv(sd, "transition2.2.0.0", xs[-1], stride=2)))]
    for m in range(4):
        xs = stage_module(sd, "stage3.%d" % m, xs, 3)
    xs = xs + [F.relu(_bn(sd, "transition3.3.0.1", _conv(sd, "transition3.3.0.0", xs[-1], stride=2)))]
    xs = stage_module(sd, "stage4.0", xs, 4)
    xs = stage_module(sd, "stage4.1", xs, 4)
    xs = stage_module(sd, "stage4.2", xs, 1)
    return _conv(sd, "final_layer", xs[0])


def hrnet_forward_chunked(sd: Dict, images, max_batch_size: int = 32) -> torch.Tensor:
    """SimpleHRNet.py:284-296: whole batch if it fits, else ``max_batch_size`` slices."""
    n = len(images)
    if n <= max_batch_size:
        return hrnet_forward(sd, images)
    outs = [hrnet_forward(sd, images[i:i + max_batch_size]) for i in range(0, n, max_batch_size)]
    return torch.cat(outs, 0)


def decode_heatmaps(heatmaps: np.ndarray, boxes: np.ndarray) -> np.ndarray:
    """SimpleHRNet.py:297-308.  heatmaps (N,J,h,w) fp32 numpy, boxes (N,4)
    ``[x1,y1,x2,y2]`` int32 (multi-person) or float32 (single-person) ->
    pts (N,J,3) fp32 ``(y, x, confidence)``.  ``np.argmax`` = first maximum in
    row-major order; coordinates are evaluated in float64 then stored as fp32."""
    n, nj, h, w = heatmaps.shape
    pts = np.empty((n, nj, 3), dtype=np.float32)
    for i in range(n):
        for j in range(nj):
            joint = heatmaps[i, j]
            pt = np.unravel_index(np.argmax(joint), (h, w))
            pts[i, j, 0] = pt[0] * 1. / h * (boxes[i][3] - boxes[i][1]) + boxes[i][1]
            pts[i, j, 1] = pt[1] * 1. / w * (boxes[i][2] - boxes[i][0]) + boxes[i][0]
            pts[i, j, 2] = joint[pt]
    return pts


def predict_crops(sd: Dict, images, boxes: np.ndarray, max_batch_size: int = 32):
    """The whole hot path (model call + decode), SimpleHRNet.py:281-308."""
    out = hrnet_forward_chunked(sd, images, max_batch_size).detach().cpu().numpy()
    return out, decode_heatmaps(out, boxes)


def cast_state_dict(sd: Dict, dtype=torch.float64) -> Dict:
    """fp64 copy: an independent cross-check of the fp32 oracle's rounding."""
    out = {}
    for k, v in sd.items():
        t = _t(sd, k)
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out


# ------------------------------------------------------------------------------------------------------------------
# Flip test-time augmentation + evaluation decode.  Restates testing/Test.py:132-140,
# misc/utils.py:9-29 (flip_tensor / flip_back), :125-151 (get_max_preds) and the post-processing loop of
# get_final_preds (:162-175).  The inverse affine (transform_preds, cv2) is out of scope.  Pinned to the reference's
# own functions through tests/golden/w32_128x96_fliptta_n3.npz.
def flip_tta_heatmaps(sd, images: torch.Tensor, flip_pairs) -> torch.Tensor:
    """``(model(x) + flip_back(model(flip(x)), pairs)) * 0.5`` -> (n,J,h,w) fp32."""
    out = hrnet_forward(sd, images)
    out_f = hrnet_forward(sd, torch.flip(images, dims=[-1]))
    out_f = torch.flip(out_f, dims=[-1]).clone()
    for a, b in flip_pairs:
        tmp = out_f[:, a].clone()
        out_f[:, a] = out_f[:, b]
        out_f[:, b] = tmp
    return (out + out_f) * 0.5


def max_preds_refined(heatmaps: np.ndarray, post_processing: bool = True):
    """``get_max_preds`` + the quarter-pixel refinement: heat-maps (n,J,h,w) -> preds (n,J,2) = (x, y), maxvals (n,J,1)."""
    n, nj, h, w = heatmaps.shape
    flat = heatmaps.reshape(n, nj, -1)
    idx = flat.argmax(-1)                       # first maximum, like torch.max
    maxvals = np.take_along_axis(flat, idx[..., None], -1).astype(np.float32)
    preds = np.stack([(idx % w).astype(np.float32), (idx // w).astype(np.float32)], -1)
    preds *= (maxvals > 0.0).astype(np.float32)
    if post_processing:
        for i in range(n):
            for j in range(nj):
                px, py = int(math.floor(preds[i, j, 0] + 0.5)), int(math.floor(preds[i, j, 1] + 0.5))
                if 1 < px < w - 1 and 1 < py < h - 1:
                    hm = heatmaps[i, j]
                    preds[i, j, 0] += np.sign(hm[py, px + 1] - hm[py, px - 1]) * np.float32(0.25)
                    preds[i, j, 1] += np.sign(hm[py + 1, px] - hm[py - 1, px]) * np.float32(0.25)
    return preds, maxvals


# ------------------------------------------------------------------------------------------------------------------
# PoseResNet: models_/poseresnet.py:108-122 restated with torch.nn.functional over the
# state_dict; Bottleneck = models_/modules.py:19-40.  Pinned by tests/golden/poseresnet50_128x96_n2.npz (the unmodified
# reference class run in the build container).
RESNET_LAYERS = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}


def _bn_eval(sd, prefix, x):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], False, 0.0, 1e-5)


def poseresnet_forward(sd, x: torch.Tensor, resnet_size: int = 50) -> torch.Tensor:
    x = F.relu(_bn_eval(sd, "bn1", F.conv2d(x, sd["conv1.weight"], None, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, blocks in enumerate(RESNET_LAYERS[resnet_size]):
        for b in range(blocks):
            p = "layer%d.%d" % (li + 1, b)
            stride = 2 if (b == 0 and li > 0) else 1
            out = F.relu(_bn_eval(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"])))
            out = F.relu(_bn_eval(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], None, stride, 1)))
            out = _bn_eval(sd, p + ".bn3", F.conv2d(out, sd[p + ".conv3.weight"]))
            res = x
            if b == 0:
                res = _bn_eval(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride))
            x = F.relu(out + res)
    for i in range(3):
        x = F.conv_transpose2d(x, sd["deconv_layers.%d.weight" % (3 * i)], None, 2, 1, 0)
        x = F.relu(_bn_eval(sd, "deconv_layers.%d" % (3 * i + 1), x))
    return F.conv2d(x, sd["final_layer.weight"], sd["final_layer.bias"])


# ------------------------------------------------------------------------------------------------------------------
# Engine-arithmetic restatement ("bf16 emulation").  The SAME graph as hrnet_forward above (models_/hrnet.py:157-189,
# :55-71; models_/modules.py:20-40, 56-72), evaluated the way the bf16 engine evaluates it, so that the bf16 HIP kernels
# can be pinned to an oracle tighter than "a few % of sigma away from the fp32 reference":
#   * every (conv, BatchNorm) pair folded exactly as hrn_load_weights folds it (simple-hrnet_amd/csrc/hrnet_mi355.cpp
#     bn_fold / load_weights): scale = gamma / sqrt(var + 1e-5) and shift = beta - mean * scale in float64,
#     W' = float32(float64(W) * scale), b' = float32(shift);
#   * round_weights: W' rounded to bf16 (round to nearest even), the head's weights likewise, biases stay fp32;
#   * round_acts: every tensor the engine writes to HBM is rounded to bf16 at that point -- conv outputs after bias +
#     residual + ReLU, the fuse sums after their ReLU, the 1x1 fuse convs at low resolution (before the upsample), the
#     stem's input pixels (the MFMA stem converts the fp32 crops); accumulation is fp32 throughout, heat-maps stay fp32.
# With both switches off it is the fp32 reference up to the re-association of the BatchNorm fold (tests/test_oracle.py
# pins that tap by tap against the reference's own forward hooks), which is what pins the emulation itself.
#
# The graph is explicit: every stored tensor is a node (named like the engine's taps, include/hrnet_mi355.h) with its
# operation and the names of its inputs, so that ONE operation can be evaluated on given inputs (``eval_node``): the
# per-op parity tests (tests/test_bf16_pin.py) feed the engine's own stored inputs of an op to the emulation of that op --
# the outputs then differ by fp32 summation order only (oneDNN here, the MFMA pipeline there): at most one bf16 ulp on
# a small fraction of the elements.  End to end the two drift apart like any two bf16 evaluations of a 100-layer net
# do (a 1-ulp flip changes what every later rounding sees), which is why the end-to-end figures are reported, not pinned.
def _bf16r(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


class EngineEmulation:
    INPUT = "input"     # the (n,3,H,W) crops
    HEAD = "heatmaps"   # final_layer output (fp32, never rounded)

    def __init__(self, sd: Dict, round_weights: bool = True, round_acts: bool = True):
        self.sd, self.rw, self.ra = sd, round_weights, round_acts
        self.graph: Dict[str, dict] = {}
        self.order: List[str] = []
        self._folded: Dict[str, tuple] = {}
        self._build()

    # ---- graph construction (mirrors HRNet.forward; no arithmetic here)
    def _node(self, name, **kw):
        self.graph[name] = kw
        self.order.append(name)
        return name

    def _conv(self, conv, bn, x, stride=1, relu=True, res=None):
        return self._node(conv, op="conv", bn=bn, x=x, stride=stride, relu=relu, res=res)

    def _bottleneck(self, p, x, has_downsample):
        o = self._conv(p + ".conv1", p + ".bn1", x)
        o = self._conv(p + ".conv2", p + ".bn2", o)
        r = x
        if has_downsample:  # (the bf16 engine computes it inside the chain kernel and rounds it like a stored tensor)
            r = self._conv(p + ".downsample.0", p + ".downsample.1", x, relu=False)
        return self._conv(p + ".conv3", p + ".bn3", o, res=r)

    def _stage(self, p, xs, nout):
        nb = len(xs)
        ys = []
        for b in range(nb):
            y = xs[b]
            for k in range(4):
                q = "%s.branches.%d.%d" % (p, b, k)
                o = self._conv(q + ".conv1", q + ".bn1", y)
                y = self._conv(q + ".conv2", q + ".bn2", o, res=y)
            ys.append(y)
        outs = []
        for i in range(nout):
            terms = []
            for j in range(nb):
                q = "%s.fuse_layers.%d.%d" % (p, i, j)
                if i == j:
                    terms.append((ys[j], 0))
                elif i < j:   # 1x1 conv + BN stored at low resolution, read upsampled by the fuse (hrnet.py:30-35)
                    terms.append((self._conv(q + ".0", q + ".1", ys[j], relu=False), j - i))
                else:         # chain of 3x3 stride-2 convs, ReLU on all but the last (hrnet.py:36-51)
                    t = ys[j]
                    for k in range(i - j):
                        t = self._conv("%s.%d.0" % (q, k), "%s.%d.1" % (q, k), t, stride=2, relu=k < i - j - 1)
                    terms.append((t, 0))
            outs.append(self._node("%s.fuse.%d" % (p, i), op="fuse", terms=terms))
        return outs

    def _build(self):
        x = self._node("stem", op="stem", x=self.INPUT)
        x = self._conv("conv2", "bn2", x, stride=2)
        for k in range(4):
            x = self._bottleneck("layer1.%d" % k, x, k == 0)
        xs = [self._conv("transition1.0.0", "transition1.0.1", x), self._conv("transition1.1.0.0", "transition1.1.0.1", x, stride=2)]
        xs = self._stage("stage2.0", xs, 2)
        xs = xs + [self._conv("transition2.2.0.0", "transition2.2.0.1", xs[-1], stride=2)]
        for m in range(4):
            xs = self._stage("stage3.%d" % m, xs, 3)
        xs = xs + [self._conv("transition3.3.0.0", "transition3.3.0.1", xs[-1], stride=2)]
        xs = self._stage("stage4.0", xs, 4)
        xs = self._stage("stage4.1", xs, 4)
        xs = self._stage("stage4.2", xs, 1)
        self._node(self.HEAD, op="head", x=xs[0])

    # ---- arithmetic
    def _fold(self, conv, bn):
        if conv not in self._folded:
            w = _t(self.sd, conv + ".weight").to(torch.float64)
            g, b = _t(self.sd, bn + ".weight").to(torch.float64), _t(self.sd, bn + ".bias").to(torch.float64)
            mu, var = _t(self.sd, bn + ".running_mean").to(torch.float64), _t(self.sd, bn + ".running_var").to(torch.float64)
            scale = g / torch.sqrt(var + 1e-5)
            shift = b - mu * scale
            wf = (w * scale.view(-1, 1, 1, 1)).to(torch.float32)
            if self.rw:
                wf = _bf16r(wf)
            self._folded[conv] = (wf, shift.to(torch.float32))
        return self._folded[conv]

    def _store(self, t):
        return _bf16r(t) if self.ra else t

    def inputs_of(self, name) -> List[str]:
        nd = self.graph[name]
        if nd["op"] == "fuse":
            return [t for t, _ in nd["terms"]]
        return [nd["x"]] + ([nd["res"]] if nd.get("res") else [])

    @torch.no_grad()
    def eval_node(self, name, vals: Dict[str, torch.Tensor], magnitude: bool = False):
        """the stored value of node ``name`` from the stored values of its inputs (``vals[input name]``).  With
        ``magnitude`` also the sum of the absolute values of everything that was added up per element (the scale fp32
        summation noise is proportional to)."""
        nd = self.graph[name]
        mag = None
        if nd["op"] == "stem":
            x = vals[nd["x"]].to(torch.float32)
            if self.ra:
                x = _bf16r(x)   # stem_mfma_kernel: the fp32 crop values become bf16 MFMA operands
            w, b = self._fold("conv1", "bn1")
            y = self._store(F.relu(F.conv2d(x, w, b, stride=2, padding=1)))
            if magnitude:
                mag = F.conv2d(x.abs(), w.abs(), b.abs(), stride=2, padding=1)
        elif nd["op"] == "conv":
            w, b = self._fold(name, nd["bn"])
            x = vals[nd["x"]]
            y = F.conv2d(x, w, b, stride=nd["stride"], padding=w.shape[-1] // 2)
            if magnitude:
                mag = F.conv2d(x.abs(), w.abs(), b.abs(), stride=nd["stride"], padding=w.shape[-1] // 2)
            if nd["res"]:
                y = y + vals[nd["res"]]
                if magnitude:
                    mag = mag + vals[nd["res"]].abs()
            if nd["relu"]:
                y = F.relu(y)
            y = self._store(y)
        elif nd["op"] == "fuse":
            acc = None
            for t, shift in nd["terms"]:   # left to right, fp32 (hrnet.py:63-66)
                v = vals[t]
                if shift:
                    v = F.interpolate(v, scale_factor=float(2 ** shift), mode="nearest")
                acc = v if acc is None else acc + v
                if magnitude:
                    mag = v.abs() if mag is None else mag + v.abs()
            y = self._store(F.relu(acc))
        else:  # head: final_layer 1x1 conv + bias (hrnet.py:155,187); bf16 weights on the MFMA head, fp32 output
            wh, bh = _t(self.sd, "final_layer.weight").to(torch.float32), _t(self.sd, "final_layer.bias").to(torch.float32)
            if self.rw:
                wh = _bf16r(wh)
            y = F.conv2d(vals[nd["x"]], wh, bh)
            if magnitude:
                mag = F.conv2d(vals[nd["x"]].abs(), wh.abs(), bh.abs())
        return (y, mag) if magnitude else y

    @torch.no_grad()
    def forward(self, images, taps=None):
        """-> heat-maps, or (heat-maps, {name: stored tensor}) for the names in ``taps`` (a set, or "all")"""
        x = images if isinstance(images, torch.Tensor) else torch.from_numpy(np.asarray(images))
        vals = {self.INPUT: x.to(torch.float32)}
        last_use = {}
        for name in self.order:
            for i in self.inputs_of(name):
                last_use[i] = name
        kept = {}
        for name in self.order:
            vals[name] = self.eval_node(name, vals)
            if taps is not None and name != self.HEAD and (taps == "all" or name in taps):
                kept[name] = vals[name]
            for i in self.inputs_of(name):   # free what nobody reads any more
                if last_use[i] == name and i in vals:
                    del vals[i]
        out = vals[self.HEAD]
        return out if taps is None else (out, kept)


def hrnet_forward_engine(sd: Dict, images, round_weights: bool = True, round_acts: bool = True, taps=None):
    """-> heat-maps (n,J,h,w) fp32, or (heat-maps, {tap name: tensor}) when ``taps`` is given."""
    return EngineEmulation(sd, round_weights, round_acts).forward(images, taps)


class PoseResNetEmulation(EngineEmulation):
    """The same engine-arithmetic restatement for the other model of the selector (models_/poseresnet.py:16-122; restated in
    fp32 by ``poseresnet_forward`` above, which is pinned to the reference fixture): 7x7 stem, max-pool, Bottleneck layers
    (stride in conv2, projection shortcut in the first block of a layer), three ConvTranspose2d(4, s2, p1) + BN + ReLU, head.
    Node names = the engine's taps: "stem", "maxpool", "<layer>.<block>.conv1..3", "<layer>.<block>.downsample.0",
    "deconv_layers.0 / 3 / 6" (the transposed convolution; its BatchNorm is deconv_layers.1 / 4 / 7)."""

    def __init__(self, sd: Dict, resnet_size: int = 50, round_weights: bool = True, round_acts: bool = True):
        self.size = resnet_size
        super().__init__(sd, round_weights, round_acts)

    def _build(self):
        x = self._node("stem", op="stem7", x=self.INPUT)
        x = self._node("maxpool", op="maxpool", x=x)
        for li, blocks in enumerate(RESNET_LAYERS[self.size]):
            for b in range(blocks):
                p = "layer%d.%d" % (li + 1, b)
                stride = 2 if (b == 0 and li > 0) else 1
                o = self._conv(p + ".conv1", p + ".bn1", x)
                o = self._conv(p + ".conv2", p + ".bn2", o, stride=stride)
                r = x
                if b == 0:
                    r = self._conv(p + ".downsample.0", p + ".downsample.1", x, stride=stride, relu=False)
                x = self._conv(p + ".conv3", p + ".bn3", o, res=r)
        for i in range(3):
            x = self._node("deconv_layers.%d" % (3 * i), op="deconv", bn="deconv_layers.%d" % (3 * i + 1), x=x)
        self._node(self.HEAD, op="head", x=x)

    @torch.no_grad()
    def eval_node(self, name, vals, magnitude: bool = False):
        nd = self.graph[name]
        if nd["op"] not in ("stem7", "maxpool", "deconv"):
            return super().eval_node(name, vals, magnitude)
        mag = None
        if nd["op"] == "stem7":
            x = vals[nd["x"]].to(torch.float32)
            if self.ra:
                x = _bf16r(x)
            w, b = self._fold("conv1", "bn1")
            y = self._store(F.relu(F.conv2d(x, w, b, stride=2, padding=3)))
            if magnitude:
                mag = F.conv2d(x.abs(), w.abs(), b.abs(), stride=2, padding=3)
        elif nd["op"] == "maxpool":
            y = F.max_pool2d(vals[nd["x"]], 3, 2, 1)   # values are stored bf16 already: the maximum of stored values
            if magnitude:
                mag = y.abs()
        else:  # ConvTranspose2d(cin, cout, 4, 2, 1) weight [ci][co][ky][kx] + BatchNorm over co + ReLU
            key = name
            if key not in self._folded:
                w = _t(self.sd, name + ".weight").to(torch.float64)
                bn = nd["bn"]
                g, b = _t(self.sd, bn + ".weight").to(torch.float64), _t(self.sd, bn + ".bias").to(torch.float64)
                mu, var = _t(self.sd, bn + ".running_mean").to(torch.float64), _t(self.sd, bn + ".running_var").to(torch.float64)
                scale = g / torch.sqrt(var + 1e-5)
                wf = (w * scale.view(1, -1, 1, 1)).to(torch.float32)
                if self.rw:
                    wf = _bf16r(wf)
                self._folded[key] = (wf, (b - mu * scale).to(torch.float32))
            w, b = self._folded[key]
            x = vals[nd["x"]]
            y = self._store(F.relu(F.conv_transpose2d(x, w, b, 2, 1, 0)))
            if magnitude:
                mag = F.conv_transpose2d(x.abs(), w.abs(), b.abs(), 2, 1, 0)
        return (y, mag) if magnitude else y
