"""The 96-cout form of the LDS-staged 3x3 kernel (simple-hrnet_amd/csrc/conv3x3_n96.inc): the BasicBlock convolutions of the
96 / 192 / 384-channel branches of HRNet-W48 (models_/modules.py:43-72).

CPU: the plan takes the form for exactly those convolutions, at every batch size (its K order differs from the other forms',
so it must not depend on the batch); that every (conv, cout tile, M tile) of the launches that now mix two forms is produced
exactly once is test_host_logic.py's test_block_maps_cover_every_tile_exactly_once, the weight image test_fold_and_pack.
GPU: every output element of the kernel against a naive convolution on the same flat padded tensors (tools/c3n_test.hip:
ragged tiles, 128-pixel tiles, with / without residual, one / several tiles per block, the three channel widths at 256
crops); the whole net with the form on and off agrees within the bf16 bound; the form does not depend on the batch a crop
arrives in."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT, load_pkg, state_dict_np

pkg = load_pkg()


def _plan(c=48, h=384, w=288, mb=4):
    return pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=mb, device=-1)


def test_plan_takes_the_form_for_the_wide_branches_only():
    net = _plan()
    infos = net.conv_infos()
    bb = [i for i in infos if b".branches." in i.name]
    assert len(bb) == 208
    for i in bb:
        if i.cin == 48:
            assert i.algo in (1, 2) and i.ks == 48 and i.nr == 3, i.name          # the 48-channel branch: (48, 3) / fused
        else:
            assert i.algo == 3 and i.ks == 32 and i.nr == 6 and i.cin % 96 == 0, i.name
    # nothing else takes it (transitions, layer1, fuse layers)
    assert all(i.algo != 3 for i in infos if b".branches." not in i.name)
    net.close()
    # widths whose branches are not multiples of 96 never do
    net = _plan(c=32, h=128, w=96)
    assert all(i.algo != 3 for i in net.conv_infos())
    net.close()


def test_form_can_be_switched_off(monkeypatch):
    monkeypatch.setenv("HRN_DISABLE_N96", "1")
    net = _plan(h=128, w=96)
    assert all(i.algo != 3 for i in net.conv_infos())
    net.close()


@pytest.mark.gpu
def test_kernel_against_a_naive_convolution(tmp_path):
    """tools/c3n_test.hip, built here with hipcc: every output element of 12 cases, then the guard rows"""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "c3n_test")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-o", exe, os.path.join(ROOT, "tools", "c3n_test.hip")])
    out = subprocess.run([exe, "64"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "all shapes OK" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.gpu
def test_net_with_and_without_the_form_agree_within_the_bf16_bound(monkeypatch):
    """same weights, same crops: the 96-cout form accumulates K in another order than the (48, 3) form, so the two nets
    differ by bf16 roundings only -- well inside the bound both keep against the fp32 oracle"""
    c, h, w, n = 48, 256, 192, 6
    crops = torch.from_numpy(pkg.synth_crops(n, h, w, seed=61)).cuda()
    boxes = pkg.synth_boxes(n, seed=62)
    outs = []
    for off in (False, True):
        monkeypatch.delenv("HRN_DISABLE_N96", raising=False)
        if off:
            monkeypatch.setenv("HRN_DISABLE_N96", "1")
        net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=4, device=0).load_state_dict(state_dict_np(c))
        assert any(i.algo == 3 for i in net.conv_infos()) != off
        hm, _ = net.predict_crops(crops, boxes, return_heatmaps=True)
        outs.append(hm.cpu().numpy())
        net.close()
    sigma = outs[1].std()
    assert np.isfinite(outs[0]).all()
    assert np.abs(outs[0] - outs[1]).max() < 0.08 * sigma + 0.05, (np.abs(outs[0] - outs[1]).max(), sigma)


@pytest.mark.gpu
def test_form_does_not_depend_on_the_batch():
    """one crop alone (128-pixel tiles, MR = 1) == the same crop inside a 40-crop call (512-pixel tiles, MR = 4), bit for bit"""
    c, h, w, n = 48, 256, 192, 40
    crops = torch.from_numpy(pkg.synth_crops(n, h, w, seed=71)).cuda()
    boxes = pkg.synth_boxes(n, seed=72)
    net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=n, device=0).load_state_dict(state_dict_np(c))
    hm_all, pts_all = net.predict_crops(crops, boxes, return_heatmaps=True)
    for k in (0, 17, 39):
        hm1, pts1 = net.predict_crops(crops[k:k + 1], boxes[k:k + 1], return_heatmaps=True)
        assert torch.equal(hm1[0], hm_all[k]) and torch.equal(pts1[0], pts_all[k])
    net.close()
