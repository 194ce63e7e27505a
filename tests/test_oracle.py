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


# ------------------------------------------------------------------------------------------------------------------
# The engine-arithmetic restatement (oracle/hrnet_torch_oracle.py: EngineEmulation) -- the oracle the bf16 HIP kernels
# are pinned to (tests/test_bf16_pin.py).  It is itself pinned here: with the roundings switched off it must be the
# reference, tap by tap, on the fixtures the reference's own forward hooks produced (tests/golden/make_golden.py taps).
def _tap_index(size, k=256):
    return np.unique(np.linspace(0, size - 1, min(size, k)).astype(np.int64))


@pytest.mark.parametrize("name", ["w32_64x64_taps_n1", "w48_64x64_taps_n1"])
def test_engine_emulation_without_rounding_is_the_reference_tap_by_tap(name):
    g = golden(name)
    c = int(g["c"])
    synth = load_pkg("synth")
    sd = synth.to_torch_state_dict(state_dict_np(c, int(g["weight_seed"])))
    x = torch.from_numpy(synth.synth_crops(int(g["n"]), int(g["h"]), int(g["w"]), seed=int(g["crop_seed"])))
    hm, taps = T.hrnet_forward_engine(sd, x, round_weights=False, round_acts=False, taps="all")
    np.testing.assert_allclose(hm.numpy(), g["heatmaps"], rtol=0, atol=5e-5)
    names = [str(s) for s in g["names"]]
    assert sorted(taps) == names                           # same tensors under the same names
    for k, t in enumerate(names):
        a = taps[t].numpy()
        assert list(a.shape) == list(g["shapes"][k]), t
        flat = a.ravel()
        want = g["samples"][g["offsets"][k]:g["offsets"][k + 1]]
        scale = max(1.0, float(np.abs(want).max()))
        # the BatchNorm fold re-associates (x*w)*s + b as x*(w*s) + b: fp32 noise only
        np.testing.assert_allclose(flat[_tap_index(flat.size)], want, rtol=0, atol=2e-5 * scale, err_msg=t)
        s, sa = float(flat.astype(np.float64).sum()), float(np.abs(flat.astype(np.float64)).sum())
        assert abs(s - g["sums"][k][0]) <= 2e-6 * g["sums"][k][1] + 1e-6, t
        assert abs(sa - g["sums"][k][1]) <= 2e-6 * g["sums"][k][1] + 1e-6, t


def test_engine_emulation_rounding_behaviour():
    """What the switches do: every tap of the bf16 emulation is bf16-representable; weights-only rounding moves the result
    less than full emulation; the emulated bf16 error against fp32 is the few-%-of-sigma the engine shows."""
    g = golden("w32_64x64_n2")
    synth = load_pkg("synth")
    sd = synth.to_torch_state_dict(state_dict_np(32))
    x = torch.from_numpy(_crops(g))
    ref = torch.from_numpy(g["heatmaps"])
    full, taps = T.hrnet_forward_engine(sd, x, taps="all")
    for name, t in taps.items():
        assert torch.equal(t, t.to(torch.bfloat16).to(torch.float32)), name
    wonly = T.hrnet_forward_engine(sd, x, round_weights=True, round_acts=False)
    e_full, e_w = float((full - ref).abs().max()), float((wonly - ref).abs().max())
    sigma = float(ref.std())
    assert 0 < e_w < e_full < 0.1 * sigma, (e_w, e_full, sigma)


def test_plan_tap_names_are_the_emulation_tap_names():
    """hrn_forward_tap's names (include/hrnet_mi355.h) == the emulation's, per plan variant; the only tensors without a
    tap are the ones the plan keeps on-chip."""
    pkg = load_pkg()
    g = golden("w48_64x64_taps_n1")
    emu_names = set(str(s) for s in g["names"])
    for dtype in ("bf16", "fp32"):
        net = pkg.NativeHRNet(48, 17, (64, 64), dtype, max_batch=2, device=-1)
        infos = net.tap_infos()
        names = [t.name.decode() for t in infos]
        assert len(names) == len(set(names))
        missing = emu_names - set(names)
        # bf16: the projection shortcut of layer1.0 and (round 5) the 3x3 convs of layer1.1-3 are computed inside the chain kernel
        assert missing == ({"layer1.0.downsample.0", "layer1.1.conv2", "layer1.2.conv2", "layer1.3.conv2"} if dtype == "bf16" else set()), missing
        assert set(names) <= emu_names
        shapes = {str(s): tuple(sh[1:]) for s, sh in zip(g["names"], g["shapes"])}
        for t in infos:
            assert (t.c, t.h, t.w) == shapes[t.name.decode()], t.name
        net.close()
