"""The oracle (oracle/) against the committed outputs of the reference itself
(tests/golden/*.npz, made by tests/golden/make_golden.py) -- runs without a GPU."""
import numpy as np
import pytest
import torch

from conftest import golden, load_pkg, state_dict_np
from oracle import hrnet_c_oracle as C
from oracle import hrnet_torch_oracle as T

HEATMAP_CASES = ["w32_64x64_n2", "w48_64x64_n2", "w32_256x192_n2"]
PREDICT_CASES = ["cfg1_w32_256x192_predict_multi", "w32_128x96_predict_single", "w48_128x96_predict_batch5",
                 "w32_128x96_predict_batch_multi"]


def _crops(g):
    if "crops" in g:
        return g["crops"]
    return load_pkg("synth").synth_crops(int(g["n"]), int(g["h"]), int(g["w"]))


@pytest.mark.parametrize("name", HEATMAP_CASES + PREDICT_CASES)
def test_torch_oracle_reproduces_reference_bits(name):
    g = golden(name)
    sd = load_pkg("synth").to_torch_state_dict(state_dict_np(int(g["c"]), int(g["weight_seed"])))
    # the reference ran the model in max_batch_size chunks for the batch5 case (SimpleHRNet.py:423-429)
    mbs = 2 if name == "w48_128x96_predict_batch5" else 32
    hm, pts = T.predict_crops(sd, torch.from_numpy(_crops(g)), g["boxes"], max_batch_size=mbs)
    # same torch build, same ops, same order -> bit-exact on CPU; tolerance left for thread-count effects
    np.testing.assert_allclose(hm, g["heatmaps"], rtol=0, atol=2e-5)
    assert np.array_equal(hm.reshape(*hm.shape[:2], -1).argmax(-1),
                          g["heatmaps"].reshape(*hm.shape[:2], -1).argmax(-1))
    # batch single-person mode returns (n,1,J,3) (SimpleHRNet.py:475): same numbers, extra axis
    np.testing.assert_allclose(pts, g["pts"].reshape(pts.shape), rtol=0, atol=2e-5)
    assert pts.dtype == np.float32


@pytest.mark.parametrize("name", ["w32_64x64_n2", "w48_64x64_n2", "w32_128x96_predict_single"])
def test_c_oracle_matches_reference(name):
    g = golden(name)
    c = int(g["c"])
    sd = state_dict_np(c, int(g["weight_seed"]))
    hm = C.hrnet_forward(sd, _crops(g), c)
    np.testing.assert_allclose(hm, g["heatmaps"], rtol=0, atol=5e-5)
    pts = C.decode_heatmaps(hm, g["boxes"])
    # argmax must agree (gaps on these weights are >= 2e-4), hence coordinates are identical
    np.testing.assert_array_equal(pts[..., :2], g["pts"][..., :2])
    np.testing.assert_allclose(pts[..., 2], g["pts"][..., 2], rtol=0, atol=5e-5)


def test_decode_tie_break_and_dtypes():
    """np.argmax takes the FIRST maximum (row-major); float64 evaluation, fp32 store (SimpleHRNet.py:302-308)."""
    hm = np.zeros((1, 2, 4, 6), np.float32)
    hm[0, 0, 1, 2] = hm[0, 0, 3, 5] = 7.0   # tie -> (1,2)
    hm[0, 1] = -1.0                          # all equal -> (0,0)
    bi = np.array([[-13, 7, 1000, 901]], np.int32)
    bf = np.array([[0, 0, 333.3, 777.7]], np.float32)
    for boxes in (bi, bf):
        a, b = T.decode_heatmaps(hm, boxes), C.decode_heatmaps(hm, boxes)
        np.testing.assert_array_equal(a, b)
        assert a[0, 0, 0] == np.float32(1 * 1. / 4 * (boxes[0][3] - boxes[0][1]) + boxes[0][1])
        assert a[0, 0, 1] == np.float32(2 * 1. / 6 * (boxes[0][2] - boxes[0][0]) + boxes[0][0])
        assert a[0, 1, 0] == np.float32(boxes[0][1]) and a[0, 1, 1] == np.float32(boxes[0][0])
        assert a[0, 0, 2] == 7.0 and a[0, 1, 2] == -1.0


def test_fp64_crosscheck_small():
    """independent rounding check: fp32 oracle vs the same graph in fp64."""
    g = golden("w32_64x64_n2")
    sd = load_pkg("synth").to_torch_state_dict(state_dict_np(32))
    x = torch.from_numpy(_crops(g))
    y32 = T.hrnet_forward(sd, x).numpy()
    y64 = T.hrnet_forward(T.cast_state_dict(sd), x.double()).numpy()
    assert np.abs(y32 - y64).max() < 5e-5
