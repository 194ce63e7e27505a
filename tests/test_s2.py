"""Stride-2 slab kernel (simple-hrnet_amd/csrc/conv_s2.hip; reference: the fuse-down chains of models_/hrnet.py:36-51).

CPU: the block map covers every (problem, tile) exactly once at any batch size, the parts cover every output channel of
every member convolution exactly once, a tile's slab fits its LDS buffer.  GPU: with the kernel on / off the whole net is
BIT-IDENTICAL (same K order and arithmetic as the generic kernel it replaces) on every geometry the row tiling meets --
full tiles, partial last tiles, one tile per image, pad columns -- and on the batch-256 path; the kernel against the
emulation directly is tests/test_bf16_pin.py."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_pkg, state_dict_np


def _plan(net, group, n):
    blocks = (ctypes.c_int32 * (3 * 65536))()
    parts = (ctypes.c_int32 * (5 * 256))()
    act = ctypes.c_int32()
    r = net._lib.hrn_plan_s2_map(net._h, group, n, blocks, 65536, parts, 256, ctypes.byref(act))
    if r < 0:
        return None
    nb, npart = r & 0xfffff, r >> 20
    return np.array(blocks[:3 * nb]).reshape(-1, 3), np.array(parts[:5 * npart]).reshape(-1, 5), act.value


@pytest.mark.parametrize("c,h,w,mb", [(48, 384, 288, 256), (48, 256, 192, 7), (48, 128, 96, 33), (32, 256, 192, 256), (32, 128, 96, 5), (96, 64, 64, 3)])
def test_s2_block_map_covers_every_tile_once(c, h, w, mb):
    pkg = load_pkg()
    net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=mb, device=-1)
    infos = net.conv_infos()
    s2 = {i for i, ci in enumerate(infos) if ci.algo == 4}
    for ci in s2:
        assert infos[ci].ksize == 3 and infos[ci].stride == 2 and infos[ci].cin in (32, 48, 64)
    if c == 96:                # widths 96 / 192 / 384 / 768: only the 64 -> 64 stem conv has a supported input width
        assert {infos[i].name.decode() for i in s2} <= {"conv2"}
        if not s2:
            net.close()
            return
    else:
        fuse = {i for i in s2 if b"fuse_layers" in infos[i].name}
        assert len(fuse) == (25 if c == 48 else 25 + 10)       # W48: branch 0 (cin 48); W32: branches 0 and 1 (cin 32, 64)
        assert s2 - fuse <= {i for i, ci in enumerate(infos) if ci.name in (b"conv2", b"transition2.2.0.0")}
    seen_parts = set()
    g = 0
    while True:
        for n in sorted({1, 2, mb // 2 + 1, mb}):
            pl = _plan(net, g, n)
            if pl is None:
                break
            blocks, parts, active = pl
            tiles = {}
            for prob, conv, tile48, rows, tpi in parts:
                ci = infos[conv]
                wop = ci.out_w + 1
                cap = {32: 1248, 48: 832, 64: 608}[ci.cin]           # input pixels one slab buffer holds (kernels.h: s2_slot_capacity)
                assert (2 * rows + 1) * 2 * wop <= cap and 1 <= rows <= ci.out_h
                assert tpi == -(-ci.out_h // rows)
                tiles[prob] = n * tpi
                if n == mb:
                    assert (conv, tile48) not in seen_parts
                    seen_parts.add((conv, tile48))
            assert max(np.bincount(parts[:, 0])) <= 8
            cover = {p: np.zeros(t, np.int32) for p, t in tiles.items()}
            for prob, cnt, t0 in blocks:
                assert cnt >= 1
                cover[prob][t0:t0 + cnt] += 1
            for p, cv in cover.items():
                assert (cv == 1).all(), (g, n, p)
            assert active == (sum(tiles.values()) >= 256)
        else:
            g += 1
            continue
        break
    assert g >= 1
    # every cout tile (48 couts for cin = 48, 32 otherwise) of every member convolution is somebody's part
    assert seen_parts == {(ci, t) for ci in s2 for t in range(infos[ci].cout // (48 if infos[ci].cin == 48 else 32))}
    net.close()


@pytest.mark.gpu
@pytest.mark.parametrize("c,h,w,n", [(48, 384, 288, 3), (48, 256, 192, 5), (48, 128, 96, 4), (48, 64, 64, 2), (48, 320, 224, 2), (48, 512, 384, 2), (48, 96, 160, 3),
                                     (48, 32, 32, 7), (32, 256, 192, 5), (32, 384, 288, 2), (32, 128, 96, 4), (32, 64, 64, 3), (64, 128, 96, 2)])
def test_s2_kernel_on_off_is_bit_identical(monkeypatch, c, h, w, n):
    """HRN_S2_MIN_TILES=1 forces the slab kernel at any batch size; HRN_DISABLE_S2 routes the same convolutions to the
    generic kernel.  Same K order, same MFMA operand layout, bias added last in both: identical bits."""
    pkg = load_pkg()
    x = torch.from_numpy(pkg.synth_crops(n, h, w, seed=31)).cuda()
    out = {}
    for tag in ("on", "off"):
        monkeypatch.delenv("HRN_DISABLE_S2", raising=False)
        monkeypatch.setenv("HRN_S2_MIN_TILES", "1")
        if tag == "off":
            monkeypatch.setenv("HRN_DISABLE_S2", "1")
        net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=n, device=0).load_state_dict(state_dict_np(c))
        assert (sum(i.algo == 4 for i in net.conv_infos()) >= 25) == (tag == "on")
        out[tag] = net(x).cpu().numpy()
        # the member convolutions themselves, before anything downstream could hide a difference
        taps = [t.name.decode() for t in net.tap_infos() if ".fuse_layers." in t.name.decode()]
        out[tag + "_taps"] = {t: net.forward_tap(x, t).cpu().numpy() for t in taps if t.count(".") >= 5}
        net.close()
    for t, v in out["on_taps"].items():
        np.testing.assert_array_equal(v, out["off_taps"][t], err_msg=t)
    np.testing.assert_array_equal(out["on"], out["off"])


@pytest.mark.gpu
def test_s2_kernel_batch256_path_is_bit_identical_and_batch_independent(monkeypatch):
    """the timed configuration (256 crops: runs of several tiles per block, all three row tilings) on / off, and the same
    crops in a small call that takes the generic fallback"""
    pkg = load_pkg()
    c, h, w, n = 48, 384, 288, 256
    g = torch.Generator(device="cuda").manual_seed(77)
    x = torch.randn((n, 3, h, w), generator=g, device="cuda", dtype=torch.float32)
    monkeypatch.delenv("HRN_DISABLE_S2", raising=False)
    monkeypatch.delenv("HRN_S2_MIN_TILES", raising=False)
    net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=n, device=0).load_state_dict(state_dict_np(c))
    on = net(x).cpu().numpy()
    small = net(x[100:103].contiguous()).cpu().numpy()          # 3 crops: generic fallback of the same plan
    net.close()
    monkeypatch.setenv("HRN_DISABLE_S2", "1")
    net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=n, device=0).load_state_dict(state_dict_np(c))
    off = net(x).cpu().numpy()
    net.close()
    np.testing.assert_array_equal(on, off)
    np.testing.assert_array_equal(on[100:103], small)
