"""Compact enumeration of the M dimension in the 96-cout form of the BasicBlock kernel (simple-hrnet_amd/csrc/conv3x3_n96.inc, CP;
reference: the BasicBlock convolutions of the 192- / 384-channel branches, models_/modules.py:43-72): tiles of REAL pixels, no
matrix instruction on the pad column / pad row of the flat layout.

CPU: which convolutions take it (grids whose padding is >= 8 % of the flat pixels), the slab bound, the switch.  GPU: with it on /
off the net is BIT-IDENTICAL -- heat-maps and the member convolutions' own outputs -- at batch sizes that take 128-pixel tiles,
512-pixel tiles, a ragged last tile, several tiles per block, and on the 256-crop path; the pad positions of every tensor stay zero."""
import numpy as np
import pytest
import torch

from conftest import load_pkg, state_dict_np


def test_plan_takes_it_where_the_padding_pays(monkeypatch):
    pkg = load_pkg()
    monkeypatch.delenv("HRN_DISABLE_COMPACT", raising=False)
    net = pkg.NativeHRNet(48, 17, (384, 288), "bf16", max_batch=256, device=-1)
    infos = net.conv_infos()
    took = {(i.out_h, i.out_w) for k, i in enumerate(infos) if net.conv_compact(k)}
    assert took == {(24, 18), (12, 9)}                         # 9.1 % and 16.9 % padding; 48x36 (4.7 %) stays flat
    for k, i in enumerate(infos):
        if net.conv_compact(k):
            assert i.algo == 3 and b".branches." in i.name and i.cin in (192, 384)
        elif i.algo == 3:
            assert (i.out_h, i.out_w) == (48, 36)
    assert sum(net.conv_compact(k) for k in range(len(infos))) == 56 + 24
    net.close()
    net = pkg.NativeHRNet(48, 17, (256, 192), "bf16", max_batch=64, device=-1)
    infos = net.conv_infos()
    # (8x6: a 512-pixel tile spans 690 flat rows with its halo -- more than a slab buffer holds once the last DMA chunk is rounded up)
    assert {(i.out_h, i.out_w) for k, i in enumerate(infos) if net.conv_compact(k)} == {(16, 12)}
    net.close()
    monkeypatch.setenv("HRN_DISABLE_COMPACT", "1")
    net = pkg.NativeHRNet(48, 17, (384, 288), "bf16", max_batch=256, device=-1)
    assert not any(net.conv_compact(k) for k in range(len(net.conv_infos())))
    net.close()


def test_slab_of_a_compact_tile_fits():
    """the longest run of flat rows a tile of 512 (128) real pixels spans, + halo, against the 700 rows of a slab buffer (whole 16-row DMA chunks: 688 usable)"""
    def rows(h, w, bm, crops):
        wp, hpwp, hw = w + 1, (h + 1) * (w + 1), h * w
        flat = lambda c: (c // hw) * hpwp + ((c % hw) // w) * wp + (c % hw) % w
        m = crops * hw
        return max(flat(min(c0 + bm, m) - 1) - flat(c0) + 1 for c0 in range(0, m, bm)) + 2 * wp + 2
    for h, w in [(24, 18), (12, 9), (16, 12)]:
        assert -(-rows(h, w, 512, 256) // 16) * 16 <= 700 and rows(h, w, 128, 256) <= 700, (h, w)
    assert -(-rows(8, 6, 512, 256) // 16) * 16 > 700               # the 8x6 grid of a 256x192 net stays flat
    assert rows(12, 9, 512, 256) > 512 + 2 * 10 + 2              # (it IS longer than a flat tile's slab: pad rows of 4-5 images)


def test_last_compact_tile_stays_inside_its_buffer():
    """a last tile that holds a few real pixels still stages its whole slab -- `slab_rows` flat rows from the
    window start of its first pixel -- so the zero tail guard behind the last image must be as long as the longest slab, for EVERY
    max_batch (n == max_batch with n * h * w % 512 small is the worst case), also for 128-pixel tiles."""
    import os, re
    from conftest import ROOT
    src = open(os.path.join(ROOT, "simple-hrnet_amd", "csrc", "kernels.h")).read()
    guard = int(re.search(r"constexpr int kConvBlockRows = (\d+);", src).group(1))
    for h, w in [(24, 18), (12, 9), (16, 12)]:
        wp, hpwp, hw = w + 1, (h + 1) * (w + 1), h * w
        flat = lambda c: (c // hw) * hpwp + ((c % hw) // w) * wp + (c % hw) % w
        for bm in (512, 128):
            worst = 0
            for crops in range(1, 257):
                m = crops * hw
                slab_rows = max(flat(min(c0 + bm, m) - 1) - flat(c0) + 1 for c0 in range(0, m, bm)) + 2 * wp + 2
                c_last = (m - 1) // bm * bm
                end = flat(c_last) - wp - 1 + slab_rows            # first row past what the last tile's LDS-DMA reads
                rows_behind_image0 = crops * hpwp + wp + 1 + guard  # ctx_plan.inc: new_tensor()
                worst = max(worst, end - crops * hpwp)
                assert end <= rows_behind_image0, (h, w, bm, crops, end, rows_behind_image0)
            if bm == 512 and (h, w) != (16, 12): assert worst > wp + 1 + 512              # (the round-4 guard of 512 rows was too short: the case ADVICE found)


CASES = [(48, 384, 288, 3, 3), (48, 384, 288, 37, 37), (48, 384, 288, 147, 147), (48, 384, 288, 19, 19), (48, 384, 288, 64, 64), (48, 256, 192, 5, 5), (48, 256, 192, 64, 64), (48, 128, 96, 33, 33),
         (48, 256, 192, 250, 256), (48, 384, 288, 250, 256), (48, 320, 224, 100, 128)]


@pytest.mark.gpu
@pytest.mark.parametrize("c,h,w,n,mb", CASES)
def test_compact_on_off_is_bit_identical(monkeypatch, c, h, w, n, mb):
    pkg = load_pkg()
    x = torch.from_numpy(pkg.synth_crops(n, h, w, seed=47)).cuda()
    out = {}
    for tag in ("on", "off"):
        monkeypatch.delenv("HRN_DISABLE_COMPACT", raising=False)
        if tag == "off":
            monkeypatch.setenv("HRN_DISABLE_COMPACT", "1")
        net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=mb, device=0).load_state_dict(state_dict_np(c))
        ncomp = sum(net.conv_compact(k) for k in range(len(net.conv_infos())))
        assert (ncomp > 0) == (tag == "on")
        out[tag] = net(x).cpu().numpy()
        names = [i.name.decode() for k, i in enumerate(net.conv_infos()) if i.algo == 3 and i.out_h <= h // 16]
        pick = names[:2] + names[len(names) // 2:len(names) // 2 + 2] + names[-2:]
        out[tag + "_taps"] = {t: net.forward_tap(x, t).cpu().numpy() for t in pick}
        out[tag + "_small"] = net(x[:2].contiguous()).cpu().numpy()          # another micro-batch size on the same handle
        assert net.pad_violations() == 0
        net.close()
    for t, v in out["on_taps"].items():
        np.testing.assert_array_equal(v, out["off_taps"][t], err_msg=t)
        assert np.abs(v).max() > 0
    np.testing.assert_array_equal(out["on"], out["off"])
    np.testing.assert_array_equal(out["on_small"], out["off_small"])
    np.testing.assert_array_equal(out["on"][:2], out["on_small"])


@pytest.mark.gpu
def test_compact_batch256_path(monkeypatch):
    pkg = load_pkg()
    c, h, w, n = 48, 384, 288, 256
    g = torch.Generator(device="cuda").manual_seed(79)
    x = torch.randn((n, 3, h, w), generator=g, device="cuda", dtype=torch.float32)
    res = {}
    for tag in ("on", "off"):
        monkeypatch.delenv("HRN_DISABLE_COMPACT", raising=False)
        if tag == "off":
            monkeypatch.setenv("HRN_DISABLE_COMPACT", "1")
        net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=n, device=0).load_state_dict(state_dict_np(c))
        res[tag] = net(x).cpu().numpy()
        res[tag + "_part"] = net(x[:100].contiguous()).cpu().numpy()
        assert net.pad_violations() == 0
        net.close()
    np.testing.assert_array_equal(res["on"], res["off"])
    np.testing.assert_array_equal(res["on_part"], res["off_part"])
    np.testing.assert_array_equal(res["on"][:100], res["on_part"])
