"""The stem as one kernel (simple-hrnet_amd/csrc/stem_fused.hip; reference: conv1/bn1/relu/conv2/bn2/relu of
models_/hrnet.py:158-163).

CPU: which handles plan it, that its problem and block map cover every output row of conv2 once, that the two-launch path
stays in the plan.  GPU: with the kernel on / off the net is BIT-IDENTICAL -- conv2's own output (tap "conv2") and the heat
maps -- on every geometry (crop widths from 32 to 288, W32 / W48), for mirrored
crops (flip-TTA), at batch 1 and at batch 256; a call that taps "stem" still gets conv1's output."""
import numpy as np
import pytest
import torch

from conftest import load_pkg, state_dict_np


def test_planned_for_bf16_hrnet_only(monkeypatch):
    pkg = load_pkg()
    monkeypatch.delenv("HRN_DISABLE_STEM_FUSE", raising=False)
    for c, hw, dt, want in [(48, (384, 288), "bf16", True), (32, (256, 192), "bf16", True), (48, (384, 288), "fp32", False),
                            (48, (64, 64), "bf16", True), (48, (96, 160), "bf16", True), (32, (128, 96), "bf16", True)]:
        net = pkg.NativeHRNet(c, 17, hw, dt, max_batch=4, device=-1)
        assert net.stem_fused() == want, (c, hw, dt)
        names = [t.name.decode() for t in net.tap_infos()]
        assert "stem" in names and "conv2" in names           # the two launches stay in the plan (taps, the switch)
        net.close()
    net = pkg.NativeHRNet(50, 17, (256, 192), "bf16", max_batch=4, device=-1, model_name="PoseResNet")
    assert not net.stem_fused()
    net.close()
    monkeypatch.setenv("HRN_DISABLE_STEM_FUSE", "1")
    net = pkg.NativeHRNet(48, 17, (384, 288), "bf16", max_batch=4, device=-1)
    assert not net.stem_fused()
    n_off = net.launches_per_pass()
    net.close()
    monkeypatch.delenv("HRN_DISABLE_STEM_FUSE")
    net = pkg.NativeHRNet(48, 17, (384, 288), "bf16", max_batch=4, device=-1)
    assert net.launches_per_pass() == n_off - 1
    net.close()


def test_too_wide_crops_keep_the_two_launches():
    """conv2's slab (three virtual rows of 2 * (W / 4 + 1) slots + the bank pad) has to fit 480 slots: W <= 288"""
    pkg = load_pkg()
    for w, want in [(512, False), (320, False), (288, True), (256, True)]:
        net = pkg.NativeHRNet(32, 17, (128, w), "bf16", max_batch=2, device=-1)
        assert net.stem_fused() == want, w
        net.close()


def test_block_map_one_equal_run_per_cu():
    """every (image, output row of conv2) is somebody's tile exactly once; at most 256 blocks, all runs equal but the last"""
    import ctypes
    pkg = load_pkg()
    for c, hw, mb in [(48, (384, 288), 256), (32, (256, 192), 64), (48, (64, 64), 5)]:
        net = pkg.NativeHRNet(c, 17, hw, "bf16", max_batch=mb, device=-1)
        ho = hw[0] // 4
        for n in sorted({1, 2, 3, mb // 2 + 1, mb}):
            blocks = (ctypes.c_int32 * (3 * 65536))()
            parts = (ctypes.c_int32 * (5 * 16))()
            act = ctypes.c_int32()
            r = net._lib.hrn_plan_s2_map(net._h, -1, n, blocks, 65536, parts, 16, ctypes.byref(act))
            assert r >= 0
            nb, npart = r & 0xfffff, r >> 20
            b = np.array(blocks[:3 * nb]).reshape(-1, 3)
            p = np.array(parts[:5 * npart]).reshape(-1, 5)
            assert npart == 2 and (p[:, 3] == 1).all() and (p[:, 4] == ho).all() and sorted(p[:, 2]) == [0, 1]
            cover = np.zeros(n * ho, np.int32)
            for prob, cnt, t0 in b:
                assert prob == 0 and cnt >= 1
                cover[t0:t0 + cnt] += 1
            assert (cover == 1).all(), (c, hw, n)
            assert nb <= 256
            runs = sorted(b[:, 1])
            assert runs[-1] == -(-n * ho // 256) and runs[-1] - runs[0] <= max(1, runs[-1] - 1) and len(set(runs[1:])) <= 1, (n, runs[:3], runs[-3:])
        net.close()


GEOMS = [(48, 384, 288, 3), (48, 256, 192, 5), (32, 256, 192, 2), (48, 64, 64, 2), (48, 96, 160, 3), (32, 128, 96, 4), (48, 32, 32, 7),
         (32, 96, 288, 3), (32, 160, 224, 2), (48, 224, 32, 2)]


@pytest.mark.gpu
@pytest.mark.parametrize("c,h,w,n", GEOMS)
def test_fused_stem_on_off_is_bit_identical(monkeypatch, c, h, w, n):
    pkg = load_pkg()
    x = torch.from_numpy(pkg.synth_crops(n, h, w, seed=41)).cuda()
    out = {}
    for tag in ("on", "off"):
        monkeypatch.delenv("HRN_DISABLE_STEM_FUSE", raising=False)
        if tag == "off":
            monkeypatch.setenv("HRN_DISABLE_STEM_FUSE", "1")
        net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=n, device=0).load_state_dict(state_dict_np(c))
        assert net.stem_fused() == (tag == "on")
        out[tag] = net(x).cpu().numpy()
        out[tag + "_conv2"] = net.forward_tap(x, "conv2").cpu().numpy()
        out[tag + "_stem"] = net.forward_tap(x, "stem").cpu().numpy()      # (takes the two launches in both handles)
        out[tag + "_after"] = net(x).cpu().numpy()                           # and the fused kernel again afterwards
        assert net.pad_violations() == 0
        net.close()
    np.testing.assert_array_equal(out["on_conv2"], out["off_conv2"])
    np.testing.assert_array_equal(out["on_stem"], out["off_stem"])
    np.testing.assert_array_equal(out["on"], out["off"])
    np.testing.assert_array_equal(out["on"], out["on_after"])
    assert np.abs(out["on_conv2"]).max() > 0


@pytest.mark.gpu
def test_fused_stem_flip_tta_and_batch256(monkeypatch):
    pkg = load_pkg()
    c, h, w, n = 48, 384, 288, 256
    g = torch.Generator(device="cuda").manual_seed(78)
    x = torch.randn((n, 3, h, w), generator=g, device="cuda", dtype=torch.float32)
    res = {}
    for tag in ("on", "off"):
        monkeypatch.delenv("HRN_DISABLE_STEM_FUSE", raising=False)
        if tag == "off":
            monkeypatch.setenv("HRN_DISABLE_STEM_FUSE", "1")
        net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=n, device=0).load_state_dict(state_dict_np(c))
        res[tag] = net(x).cpu().numpy()
        res[tag + "_small"] = net(x[200:203].contiguous()).cpu().numpy()
        pairs = [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]]
        res[tag + "_tta"] = [t.cpu().numpy() for t in net.predict_flip_tta(x[:5].contiguous(), pairs)]
        net.close()
    np.testing.assert_array_equal(res["on"], res["off"])
    np.testing.assert_array_equal(res["on"][200:203], res["on_small"])
    np.testing.assert_array_equal(res["on_small"], res["off_small"])
    for a, b in zip(res["on_tta"], res["off_tta"]):          # the mirrored pass reads the crops right to left
        np.testing.assert_array_equal(a, b)
