"""The persistent work-queue form of the grouped BasicBlock launches (round 4, csrc/conv3x3_queue.inc).

CPU (plan-only handles, through hrn_plan_queue): the unit lists cover every (convolution, cout tile, M tile) of the
queued convolutions exactly once, units are long enough for the in-kernel look-ahead, the fused 48-channel BasicBlock is
dealt out as tile ranges over whole XCD rounds, small calls fall back to the per-block form.
GPU: the form on / off is BIT-IDENTICAL (it runs the per-block form's tile arithmetic instruction for instruction; only
who computes which tile when differs), at the headline shape and at sizes with ragged last tiles; with one tile and with
several tiles per unit; results do not depend on the order in which blocks happen to draw units (repeated calls).
"""
import numpy as np
import pytest
import torch

from conftest import load_pkg, state_dict_np


def _plan(net, n, reverse):
    lib = net._lib
    units, info = np.zeros((60000, 4), np.int32), np.zeros(4, np.int32)
    out, group = [], 0
    while True:
        nu = lib.hrn_plan_queue(net._h, group, n, reverse, units.ctypes.data, len(units), info.ctypes.data)
        if nu == -1:
            break
        out.append((nu, units[:max(nu, 0)].copy(), info.copy()))
        group += 1
    return out


@pytest.mark.parametrize("n,reverse", [(256, 0), (256, 1), (200, 0), (97, 1)])
def test_queue_units_cover_every_tile_once(monkeypatch, n, reverse):
    monkeypatch.setenv("HRN_QUEUE", "1")
    pkg = load_pkg()
    net = pkg.NativeHRNet(48, 17, (384, 288), "bf16", max_batch=256, device=-1)
    infos = net.conv_infos()
    plans = _plan(net, n, reverse)
    assert len(plans) == 69
    queued = 0
    for nu, units, info in plans:
        if nu == -2:
            continue
        queued += 1
        assert nu >= 2 * info[3] and info[3] == 256
        cov = {}
        for conv, nt, mt0, tiles in units:
            i = infos[conv]
            assert i.algo == 3 and i.cin == i.cout and i.cin % 96 == 0            # a 96-cout-form convolution
            mtiles = -(-n * (i.out_h + 1) * (i.out_w + 1) // 512)
            assert 0 <= nt < i.cout // 96 and tiles >= 1 and 0 <= mt0 and mt0 + tiles <= mtiles
            assert (i.cin // 32) * tiles >= 3                                      # draw / record + biases / use: one slice each
            c = cov.setdefault(int(conv), np.zeros((i.cout // 96, mtiles), np.int32))
            c[nt, mt0:mt0 + tiles] += 1
        assert all((c == 1).all() for c in cov.values())
        if info[0] >= 0:                                                           # the fused BasicBlock of this launch
            i = infos[info[0]]
            assert i.algo == 2 and info[2] == -(-n * (i.out_h + 1) * (i.out_w + 1) // 512)
            assert info[1] % 8 == 0 and 8 <= info[1] <= 256 - 8
    # stage 2 (one wide branch beside the fused one) and stages 3 / 4: every BasicBlock launch with a 96-cout-form member
    assert queued >= (64 if n >= 200 else 56)   # (at ~100 crops stage 2's single wide branch has too few units: per-block form)
    net.close()


def test_small_calls_take_the_per_block_form(monkeypatch):
    monkeypatch.setenv("HRN_QUEUE", "1")
    pkg = load_pkg()
    net = pkg.NativeHRNet(48, 17, (384, 288), "bf16", max_batch=256, device=-1)
    assert all(nu == -2 for nu, _, _ in _plan(net, 8, 0))
    net.close()
    net = pkg.NativeHRNet(32, 17, (256, 192), "fp32", max_batch=64, device=-1)     # fp32 mode: never
    assert all(nu == -2 for nu, _, _ in _plan(net, 64, 0))
    net.close()
    monkeypatch.delenv("HRN_QUEUE")                                                # the form is an option, not the default
    net = pkg.NativeHRNet(48, 17, (384, 288), "bf16", max_batch=256, device=-1)
    assert all(nu == -2 for nu, _, _ in _plan(net, 256, 0))
    net.close()


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,n,env", [(384, 288, 256, {}), (384, 288, 200, {"HRN_Q_TPB": "2"}), (256, 192, 250, {}),
                                       (384, 288, 256, {"HRN_Q_BBF_SCALE": "0.5"}), (384, 288, 130, {"HRN_Q_MIN_UNITS": "1"})])
def test_queue_form_is_bit_identical_to_the_per_block_form(monkeypatch, h, w, n, env):
    assert torch.cuda.is_available(), "GPU tests need a GPU: the HIP path has no CPU fallback"
    pkg = load_pkg()
    crops = torch.from_numpy(pkg.synth_crops(n, h, w, seed=61)).cuda()
    boxes = pkg.synth_boxes(n, seed=62)
    outs = []
    for on in (True, False, True):
        for k in ("HRN_QUEUE", "HRN_Q_TPB", "HRN_Q_BBF_SCALE", "HRN_Q_MIN_UNITS"):
            monkeypatch.delenv(k, raising=False)
        if on:
            monkeypatch.setenv("HRN_QUEUE", "1")
            for k, v in env.items():
                monkeypatch.setenv(k, v)
        net = pkg.NativeHRNet(48, 17, (h, w), "bf16", max_batch=256, device=0).load_state_dict(state_dict_np(48))
        nq = sum(1 for nu, _, _ in _plan(net, n, 0) if nu >= 0)
        assert (nq > 0) == on
        hm, pts = net.predict_crops(crops, boxes, return_heatmaps=True)
        hm2, _ = net.predict_crops(crops, boxes, return_heatmaps=True)            # draws land in another order: same bits
        assert torch.equal(hm, hm2)
        outs.append((hm.cpu().numpy(), pts.cpu().numpy()))
        net.close()
    assert np.isfinite(outs[0][0]).all()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    np.testing.assert_array_equal(outs[0][0], outs[2][0])
