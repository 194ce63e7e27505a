"""Round 6 (simple-hrnet_amd/csrc/kernels.hip, XL): the stride-2 3x3 convolutions with 96 / 192 / 256 / 384 input channels (reference:
transition1 / transition2 / transition3 and the fuse-down chains of models_/hrnet.py:36-51, 98-145) request their pixel fragments two
K chunks ahead by LDS-DMA into a per-wave LDS ring instead of loading every tap's fragments straight from L2.  Same tiles, weight image,
K order and MFMA sequence as the plain form: with the ring on / off every member convolution and the whole net are BIT-IDENTICAL.  The
form is only taken by launches large enough for 64-pixel-per-wave tiles, so the cases are full-size calls."""
import numpy as np
import pytest
import torch

from conftest import load_pkg, state_dict_np


def _eligible(infos):
    return [i for i in infos if i.algo == 0 and i.ksize == 3 and i.stride == 2 and i.cin % 32 == 0 and i.nr == 6 and not i.has_residual]


def test_xl_candidates_of_w48():
    """the plan's stride-2 3x3 convolutions outside the slab kernel are exactly the shapes the XL form is written for (CPU, plan only)"""
    pkg = load_pkg()
    net = pkg.NativeHRNet(48, 17, (384, 288), "bf16", max_batch=256, device=-1)
    infos = net.conv_infos()
    el = _eligible(infos)
    shapes = sorted({(i.cin, i.cout) for i in el})
    assert shapes == [(96, 96), (96, 192), (96, 384), (192, 384), (256, 96)], shapes
    assert len(el) == 7 + 3 + 2 + 2 + 1
    # everything else of stride 2 is the slab kernel's (48 input channels) or the 64 -> 64 stem convolution
    rest = [i for i in infos if i.ksize == 3 and i.stride == 2 and i not in el]
    assert all(i.cin in (48, 64) for i in rest)
    net.close()


@pytest.mark.gpu
@pytest.mark.parametrize("c,h,w,n", [(48, 384, 288, 256), (48, 256, 192, 256), (48, 320, 224, 200), (96, 192, 160, 192)])
def test_xl_on_off_is_bit_identical(monkeypatch, c, h, w, n):
    """(c = 96: a net whose every branch width is a multiple of 96 -- input channels 96 / 192 / 384 / 768 all take the form)"""
    pkg = load_pkg()
    g = torch.Generator(device="cuda").manual_seed(91)
    x = torch.randn((n, 3, h, w), generator=g, device="cuda", dtype=torch.float32)
    out = {}
    for tag in ("on", "off"):
        monkeypatch.delenv("HRN_DIRECT_XLDS", raising=False)
        if tag == "off":
            monkeypatch.setenv("HRN_DIRECT_XLDS", "0")
        net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=n, device=0).load_state_dict(state_dict_np(c))
        assert ("HRN_DIRECT_XLDS" in net.switches()) == (tag == "off")
        out[tag] = net(x).cpu().numpy()
        names = [i.name.decode() for i in _eligible(net.conv_infos())]
        taps = {t.name.decode() for t in net.tap_infos()}
        # the member convolutions themselves (crops from the first, a middle and the last M tiles), before anything downstream could hide a difference
        out[tag + "_taps"] = {nm: net.forward_tap(x, nm, crop0=0, ncrops=3, crop_step=(n - 1) // 2).cpu().numpy() for nm in names if nm in taps}
        assert len(out[tag + "_taps"]) >= 10
        net.close()
    for t, v in out["on_taps"].items():
        np.testing.assert_array_equal(v, out["off_taps"][t], err_msg=t)
    np.testing.assert_array_equal(out["on"], out["off"])
    assert np.isfinite(out["on"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [8, 12])
def test_small_launch_tile_rule_is_bit_identical(monkeypatch, n):
    """Round 6 (ctx_memory.inc: group_blocks): in a small launch the short convolutions (the 48-channel branch: two half-stages per tile) keep
    full tiles when 128-pixel tiles would make more than ~1.2 blocks per CU; the (48, 3) form's arithmetic does not depend on the tile size, so
    HRN_SMALL_KEEP=0 (every convolution on 128-pixel tiles, round 5's rule) gives the same bits -- as does the 256-pixel head split of small calls
    against the reference decode (tests/test_gpu_parity.py) and a 256-crop call that contains the same crops."""
    pkg = load_pkg()
    h, w = 384, 288
    g = torch.Generator(device="cuda").manual_seed(17)
    x = torch.randn((64, 3, h, w), generator=g, device="cuda", dtype=torch.float32)
    boxes = pkg.synth_boxes(64, seed=5)
    out = {}
    for tag in ("keep", "small"):
        monkeypatch.delenv("HRN_SMALL_KEEP", raising=False)
        if tag == "small":
            monkeypatch.setenv("HRN_SMALL_KEEP", "0")
        net = pkg.NativeHRNet(48, 17, (h, w), "bf16", max_batch=64, device=0).load_state_dict(state_dict_np(48))
        hm, pts = net.predict_crops(x[:n].contiguous(), boxes[:n], return_heatmaps=True)
        out[tag] = (hm.cpu().numpy(), pts.cpu().numpy())
        if tag == "keep":   # the same crops inside a 64-crop call (1024-pixel head slabs, full tiles everywhere)
            hm64, pts64 = net.predict_crops(x, boxes, return_heatmaps=True)
            out["big"] = (hm64[:n].cpu().numpy(), pts64[:n].cpu().numpy())
        net.close()
    for other in ("small", "big"):
        np.testing.assert_array_equal(out["keep"][0], out[other][0], err_msg=other)
        np.testing.assert_array_equal(out["keep"][1], out[other][1], err_msg=other)
