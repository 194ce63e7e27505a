"""PoseResNet (models_/poseresnet.py): synth spec and oracle against the reference (build
container / fixture), then the HIP engine against the fixture."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import golden, load_pkg
from oracle import hrnet_torch_oracle as T

NAME = "poseresnet50_128x96_n2"
REF = "/root/reference"


def _sd(pkg, size, seed):
    return pkg.synth_state_dict(size, 17, seed, model="PoseResNet")


def test_oracle_matches_reference_fixture():
    g = golden(NAME)
    pkg = load_pkg()
    sd = pkg.synth.to_torch_state_dict(_sd(pkg, int(g["c"]), int(g["weight_seed"])))
    with torch.no_grad():
        hm = T.poseresnet_forward(sd, torch.from_numpy(g["crops"]), int(g["c"])).numpy()
    # bit-identical in the build container; another host CPU may pick other oneDNN kernels (summation order)
    np.testing.assert_allclose(hm, g["heatmaps"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(T.decode_heatmaps(hm, g["boxes"])[..., :2], g["pts"][..., :2])


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
@pytest.mark.parametrize("size", [50, 101])
def test_synth_spec_equals_reference_state_dict(size):
    sys.path.insert(0, REF)
    try:
        from models_.poseresnet import PoseResNet
    finally:
        sys.path.remove(REF)
    pkg = load_pkg()
    ref = PoseResNet(size, 17).state_dict()
    spec = pkg.synth.poseresnet_state_spec(size, 17)
    assert [k for k, _, _ in spec] == list(ref.keys())
    assert all(tuple(ref[k].shape) == tuple(shape) for k, shape, _ in spec)


def test_plan_only_census_and_errors():
    pkg = load_pkg()
    net = pkg.NativeHRNet(50, 17, (256, 192), "bf16", max_batch=4, device=-1, model_name="PoseResNet")
    infos = net.conv_infos()
    assert len(infos) == 52 + 12                      # 16 Bottlenecks x 3 + 4 projections; 3 deconvs x 4 phases
    assert abs(net.flops_per_crop() / 1e9 - 10.853) < 0.01
    net.load_state_dict(_sd(pkg, 50, 0))
    net.close()
    with pytest.raises(ValueError):
        pkg.NativeHRNet(18, 17, (256, 192), "bf16", device=-1, model_name="PoseResNet")   # broken in the reference too
    with pytest.raises(ValueError):
        pkg.NativeHRNet(48, 17, (256, 192), "bf16", device=-1, model_name="VGG")


@pytest.mark.gpu
@pytest.mark.parametrize("mb", [1, 4])
def test_gpu_fp32_matches_reference(mb):
    g = golden(NAME)
    pkg = load_pkg()
    c, h, w = int(g["c"]), int(g["h"]), int(g["w"])
    net = pkg.NativeHRNet(c, 17, (h, w), "fp32", max_batch=mb, device=0, model_name="PoseResNet")
    net.load_state_dict(_sd(pkg, c, int(g["weight_seed"])))
    hm, pts = net.predict_crops(torch.from_numpy(g["crops"]).cuda(), g["boxes"], return_heatmaps=True)
    hm, pts = hm.cpu().numpy(), pts.cpu().numpy()
    np.testing.assert_allclose(hm, g["heatmaps"], rtol=0, atol=2e-5)      # heat-map sigma is 0.09 here
    np.testing.assert_array_equal(pts[..., :2], g["pts"][..., :2])
    np.testing.assert_array_equal(net(torch.from_numpy(g["crops"]).cuda()).cpu().numpy(), hm)
    net.close()


@pytest.mark.gpu
def test_gpu_bf16_bounded_and_other_shapes():
    g = golden(NAME)
    pkg = load_pkg()
    c, h, w = int(g["c"]), int(g["h"]), int(g["w"])
    sd = _sd(pkg, c, int(g["weight_seed"]))
    net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=4, device=0, model_name="PoseResNet").load_state_dict(sd)
    hm, pts = net.predict_crops(torch.from_numpy(g["crops"]).cuda(), g["boxes"], return_heatmaps=True)
    err = np.abs(hm.cpu().numpy() - g["heatmaps"]).max()
    assert err < 0.08 * g["heatmaps"].std() + 0.004, err
    np.testing.assert_array_equal(pts.cpu().numpy()[..., :2], T.decode_heatmaps(hm.cpu().numpy(), g["boxes"])[..., :2])
    net.close()
    # 256x192, 5 crops with micro-batch 2, against the oracle
    crops = pkg.synth_crops(5, 256, 192, seed=23)
    boxes = pkg.synth_boxes(5, seed=24)
    with torch.no_grad():
        ref = T.poseresnet_forward(pkg.synth.to_torch_state_dict(sd), torch.from_numpy(crops), c).numpy()
    net = pkg.NativeHRNet(c, 17, (256, 192), "fp32", max_batch=2, device=0, model_name="PoseResNet").load_state_dict(sd)
    hm, pts = net.predict_crops(torch.from_numpy(crops).cuda(), boxes, return_heatmaps=True)
    np.testing.assert_allclose(hm.cpu().numpy(), ref, rtol=0, atol=2e-5)
    np.testing.assert_array_equal(pts.cpu().numpy()[..., :2], T.decode_heatmaps(ref, boxes)[..., :2])
    net.close()


def test_engine_emulation_without_rounding_is_the_pinned_oracle():
    """oracle/hrnet_torch_oracle.py: PoseResNetEmulation (what the bf16 PoseResNet kernels are pinned to, operation by operation)
    with its roundings off against the reference fixture and the fp32 restatement, and its node names against the plan's taps"""
    g = golden(NAME)
    pkg = load_pkg()
    c = int(g["c"])
    sd = pkg.synth.to_torch_state_dict(_sd(pkg, c, int(g["weight_seed"])))
    hm = T.PoseResNetEmulation(sd, c, round_weights=False, round_acts=False).forward(torch.from_numpy(g["crops"])).numpy()
    np.testing.assert_allclose(hm, g["heatmaps"], rtol=0, atol=2e-6)
    emu = T.PoseResNetEmulation(sd, c)
    net = pkg.NativeHRNet(c, 17, (int(g["h"]), int(g["w"])), "bf16", max_batch=2, device=-1, model_name="PoseResNet")
    taps = {t.name.decode() for t in net.tap_infos()}
    # (on-chip: the projection shortcut of layer1.0 and, round 5, the 3x3 convs of layer1.1-2 -- PoseResNet-50 layer1 has three blocks)
    assert set(emu.order) - {emu.HEAD} - taps == {"layer1.0.downsample.0", "layer1.1.conv2", "layer1.2.conv2"} and taps <= set(emu.order)
    net.close()


@pytest.mark.gpu
@pytest.mark.parametrize("size,h,w,n", [(50, 128, 96, 2), (50, 256, 192, 33), (101, 128, 96, 1)])
def test_gpu_bf16_every_operation_meets_the_emulation(size, h, w, n):
    """the bf16 PoseResNet path -- 7x7 MFMA stem, max-pool, the Bottleneck convolutions (chain kernel in layer1, LDS-staged 3x3s,
    stride-2 1x1 projections), the four-phase transposed convolutions, the MFMA head -- operation by operation on the engine's
    own stored inputs, as tests/test_bf16_pin.py does for HRNet"""
    from test_bf16_pin import Pinner
    pkg = load_pkg()
    sd_np = _sd(pkg, size, 7)
    emu = T.PoseResNetEmulation(pkg.synth.to_torch_state_dict(sd_np), size)
    net = pkg.NativeHRNet(size, 17, (h, w), "bf16", max_batch=n, device=0, model_name="PoseResNet").load_state_dict(sd_np)
    x = torch.from_numpy(pkg.synth_crops(n, h, w, seed=29)).cuda()
    pin = Pinner(pkg, net, emu, x, crop0=0, ncrops=min(n, 2), crop_step=max(1, n - 1))
    pin.check_all()
    pin.report("PoseResNet-%d %dx%d n=%d" % (size, h, w, n))
    net.close()
