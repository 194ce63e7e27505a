"""Flip test-time augmentation + evaluation decode: the oracle against the reference's own
flip_tensor / flip_back / get_final_preds (fixture), then the HIP path against both."""
import numpy as np
import pytest
import torch

from conftest import golden, load_pkg, state_dict_np
from oracle import hrnet_torch_oracle as T

NAME = "w32_128x96_fliptta_n3"


def test_oracle_matches_reference_functions():
    g = golden(NAME)
    pkg = load_pkg()
    sd = pkg.synth.to_torch_state_dict(state_dict_np(int(g["c"]), int(g["weight_seed"])))
    hm = T.flip_tta_heatmaps(sd, torch.from_numpy(g["crops"]), g["flip_pairs"].tolist()).numpy()
    np.testing.assert_allclose(hm, g["heatmaps"], rtol=0, atol=1e-6)   # bit-identical in the build container
    preds, maxvals = T.max_preds_refined(g["heatmaps"], True)
    np.testing.assert_array_equal(preds, g["preds"])
    np.testing.assert_array_equal(maxvals, g["maxvals"])
    np.testing.assert_array_equal(T.max_preds_refined(g["heatmaps"], False)[0], g["preds_nopost"])
    assert np.abs(g["preds"] - g["preds_nopost"]).max() == 0.25      # the refinement is exercised


@pytest.mark.gpu
@pytest.mark.parametrize("mb", [2, 8])
def test_gpu_flip_tta_fp32_matches_reference(mb):
    g = golden(NAME)
    pkg = load_pkg()
    c, h, w = int(g["c"]), int(g["h"]), int(g["w"])
    net = pkg.NativeHRNet(c, 17, (h, w), "fp32", max_batch=mb, device=0).load_state_dict(state_dict_np(c, int(g["weight_seed"])))
    hm, preds, maxvals = net.predict_flip_tta(torch.from_numpy(g["crops"]).cuda(), g["flip_pairs"])
    hm, preds, maxvals = hm.cpu().numpy(), preds.cpu().numpy(), maxvals.cpu().numpy()
    np.testing.assert_allclose(hm, g["heatmaps"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(maxvals, g["maxvals"], rtol=0, atol=2e-4)
    # the decode of OUR averaged maps is exactly the reference's decode of them ...
    ref_preds, ref_max = T.max_preds_refined(hm, True)
    np.testing.assert_array_equal(preds, ref_preds)
    np.testing.assert_array_equal(maxvals, ref_max)
    # ... and equals the fixture wherever the +-0.25 sign is decided by more than the fp32 summation noise
    same = np.abs(preds - g["preds"]) < 1e-6
    assert same.mean() > 0.97 and np.abs(preds - g["preds"]).max() <= 0.5
    raw = net.predict_flip_tta(torch.from_numpy(g["crops"]).cuda(), g["flip_pairs"], post_processing=False)[1].cpu().numpy()
    np.testing.assert_array_equal(raw, g["preds_nopost"])
    # no pairs: plain average with the mirrored pass; empty batch
    assert tuple(net.predict_flip_tta(torch.zeros((0, 3, h, w)).cuda(), [])[1].shape) == (0, 17, 2)
    net.close()


@pytest.mark.gpu
def test_gpu_flip_tta_bf16_bounded():
    g = golden(NAME)
    pkg = load_pkg()
    c, h, w = int(g["c"]), int(g["h"]), int(g["w"])
    net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=4, device=0).load_state_dict(state_dict_np(c, int(g["weight_seed"])))
    hm, preds, maxvals = net.predict_flip_tta(torch.from_numpy(g["crops"]).cuda(), g["flip_pairs"])
    hm = hm.cpu().numpy()
    assert np.abs(hm - g["heatmaps"]).max() < 0.05 * g["heatmaps"].std() + 0.05
    ref_preds, _ = T.max_preds_refined(hm, True)
    np.testing.assert_array_equal(preds.cpu().numpy(), ref_preds)
    net.close()
