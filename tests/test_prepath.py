"""Crop pre-path (SimpleHRNet.py:236-278).

CPU: the numpy restatement of Pillow's resampler against the real ``PIL.Image.resize`` (bit-exact), the box
arithmetic, the committed fixture.  GPU: ``hrn_preprocess_frame`` against the oracle / the fixture, bit-exact, and the
whole frame -> joints chain."""
import numpy as np
import pytest
import torch

from conftest import golden, load_pkg, state_dict_np
from oracle import prepath_oracle as P

SHAPES = [(37, 29, 16, 12), (200, 150, 128, 96), (64, 48, 64, 48), (50, 300, 64, 48), (411, 123, 64, 48),
          (64, 48, 128, 96), (97, 288, 384, 288), (1000, 20, 32, 24), (33, 97, 96, 64), (17, 48, 64, 48)]


@pytest.mark.parametrize("h,w,oh,ow", SHAPES)
def test_resampler_restatement_equals_pillow(h, w, oh, ow):
    PIL = pytest.importorskip("PIL")
    img = np.random.default_rng(h * 1000 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    np.testing.assert_array_equal(P.pil_bilinear_u8(img, oh, ow), P.pil_resize(img, oh, ow))


def test_resampler_restatement_equals_pillow_on_random_sizes():
    """property test (hypothesis): any input / output size, up- and down-scaling mixed per axis, constant and noisy images"""
    pytest.importorskip("PIL")
    hyp = pytest.importorskip("hypothesis")
    st = hyp.strategies

    @hyp.settings(max_examples=60, deadline=None)
    @hyp.given(h=st.integers(1, 160), w=st.integers(1, 160), oh=st.integers(1, 96), ow=st.integers(1, 96), seed=st.integers(0, 2 ** 31),
               flat=st.booleans())
    def check(h, w, oh, ow, seed, flat):
        rng = np.random.default_rng(seed)
        img = np.full((h, w, 3), rng.integers(0, 256), np.uint8) if flat else rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        np.testing.assert_array_equal(P.pil_bilinear_u8(img, oh, ow), P.pil_resize(img, oh, ow))

    check()


def test_box_arithmetic():
    # 384x288: target aspect h/w = 4/3
    # tall enough already (cf < 1): pad x.  box 80 x 180 -> cf = 4/3 * 80/180 = 0.5926 -> length = round(80/cf) = 135
    sl, new, pad = P.crop_box([30.4, 20.5, 110.5, 200.49], 384, 288)
    assert sl == (30, 20, 110, 200)            # round-half-even: 20.5 -> 20, 110.5 -> 110
    assert new == (70 - 67, 20, 70 + 67, 200) and pad == (0, 0, 27, 27)
    # wide (cf > 1): pad y
    sl, new, pad = P.crop_box([100.5, 50.5, 300.2, 120.7], 384, 288)
    assert sl == (100, 50, 300, 121)
    length = int(round(71 * (384 / 288 * 200 / 71)))
    assert new == (100, 85 - length // 2, 300, 85 + length // 2) and pad == (50 - new[1], new[3] - 121, 0, 0)
    # exact aspect: untouched
    assert P.crop_box([10, 10, 58, 74], 64, 48) == ((10, 10, 58, 74), (10, 10, 58, 74), (0, 0, 0, 0))


def test_oracle_matches_fixture():
    g = golden("prepath_240x320_to_96x64")
    images, boxes = P.prepath(g["frame"], g["dets"], int(g["h"]), int(g["w"]))
    np.testing.assert_array_equal(boxes, g["boxes"])
    np.testing.assert_array_equal(images, g["images"])     # restated resampler == Pillow's, bit for bit


# ------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def pkg():
    p = load_pkg()
    return p


def _frame(hf, wf, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:hf, 0:wf]
    f = np.stack([(xx * 255 // wf), (yy * 255 // hf), ((xx + yy) * 255 // (hf + wf))], -1).astype(np.int32)
    return np.clip(f + rng.integers(-50, 51, f.shape), 0, 255).astype(np.uint8)


@pytest.mark.gpu
def test_gpu_prepath_matches_fixture_bit_exact(pkg):
    g = golden("prepath_240x320_to_96x64")
    net = pkg.NativeHRNet(32, 17, (int(g["h"]), int(g["w"])), "fp32", max_batch=4, device=0)
    images, boxes, boxes_dev = net.preprocess_frame(g["frame"], g["dets"])
    np.testing.assert_array_equal(boxes, g["boxes"])
    np.testing.assert_array_equal(boxes_dev.cpu().numpy(), g["boxes"])
    np.testing.assert_array_equal(images.cpu().numpy(), g["images"])
    net.close()


def test_box_arithmetic_clamped_variant():
    # SimpleHRNet.py:396-407: enlarge the short side around its centre, clamp to the frame
    assert P.crop_box_clamped([100.5, 50.5, 300.2, 120.7], 384, 288, 240, 320) == (100, 0, 300, 218)   # y grows, top clamps
    assert P.crop_box_clamped([30.4, 20.5, 110.5, 200.49], 384, 288, 240, 320) == (3, 20, 137, 200)     # x grows
    assert P.crop_box_clamped([300, 10, 318, 200], 384, 288, 240, 320)[2] == 320                         # right edge clamps
    assert P.crop_box_clamped([10, 10, 58, 74], 64, 48, 240, 320) == (10, 10, 58, 74)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["pad", "clamp"])
@pytest.mark.parametrize("res,hf,wf,seed", [((384, 288), 720, 1280, 1), ((256, 192), 480, 640, 2), ((64, 96), 333, 517, 3)])
def test_gpu_prepath_random_boxes_vs_oracle(pkg, res, hf, wf, seed, variant):
    rng = np.random.default_rng(seed)
    frame = _frame(hf, wf, seed)
    dets = []
    for _ in range(12):  # random people: any aspect, some touching the frame edges, some tiny, some huge
        bw, bh = rng.integers(6, wf), rng.integers(6, hf)
        x1, y1 = rng.uniform(0, wf - bw), rng.uniform(0, hf - bh)
        dets.append([x1, y1, x1 + bw + rng.uniform(-0.49, 0.49), y1 + bh + rng.uniform(-0.49, 0.49), 0.9, 0.9, 0])
    dets += [[0, 0, wf, hf, 1, 1, 0], [0.5, 1.5, 2.5, 3.5, 1, 1, 0]]      # whole frame; ties in the rounding
    dets = np.asarray(dets, np.float32)
    dets[:, 2] = np.minimum(dets[:, 2], wf)
    dets[:, 3] = np.minimum(dets[:, 3], hf)
    ref = P.prepath if variant == "pad" else P.prepath_clamped
    ref_images, ref_boxes = ref(frame, dets, res[0], res[1])
    net = pkg.NativeHRNet(32, 17, res, "bf16", max_batch=4, device=0)
    images, boxes, _ = net.preprocess_frame(torch.from_numpy(frame), dets, variant)
    np.testing.assert_array_equal(boxes, ref_boxes)
    np.testing.assert_array_equal(images.cpu().numpy(), ref_images)
    # and again (scratch buffers are reused), with fewer people
    images2, boxes2, _ = net.preprocess_frame(torch.from_numpy(frame).cuda(), dets[:3], variant)
    np.testing.assert_array_equal(images2.cpu().numpy(), ref_images[:3])
    net.close()


@pytest.mark.gpu
def test_gpu_frame_to_joints_and_errors(pkg):
    from oracle import hrnet_torch_oracle as T
    c, res = 32, (128, 96)
    frame = _frame(360, 480, 9)
    dets = np.asarray([[40.2, 30.7, 200.1, 330.3, .9, .9, 0], [250.5, 100.5, 460.4, 200.6, .8, .8, 0]], np.float32)
    sd = state_dict_np(c, 4)
    net = pkg.NativeHRNet(c, 17, res, "fp32", max_batch=4, device=0).load_state_dict(sd)
    boxes, pts, hm = net.predict_frame(frame, dets, return_heatmaps=True)
    ref_images, ref_boxes = P.prepath(frame, dets, *res)
    ref_hm, ref_pts = T.predict_crops(pkg.synth.to_torch_state_dict(sd), torch.from_numpy(ref_images), ref_boxes)
    np.testing.assert_array_equal(boxes, ref_boxes)
    np.testing.assert_allclose(hm.cpu().numpy(), ref_hm, rtol=0, atol=2e-4)
    np.testing.assert_array_equal(pts.cpu().numpy()[..., :2], ref_pts[..., :2])
    # no people: empty results, no launch
    b0, p0 = net.predict_frame(frame, np.zeros((0, 7), np.float32))
    assert b0.shape == (0, 4) and tuple(p0.shape) == (0, 17, 3)
    # a box outside the frame / degenerate: loud failure (the reference would wrap around or divide by zero)
    for bad in ([[-5, 10, 50, 100]], [[10, 10, 10, 100]], [[500, 10, 600, 100]]):
        with pytest.raises((ValueError, RuntimeError)):
            net.preprocess_frame(frame, np.asarray(bad, np.float32))
    net.close()


# ------------------------------------------------------------- pinned to the unmodified reference predict()
# tests/golden/make_golden.py ran SimpleHRNet.predict() itself (its box / pad / slice code, :236-278 and :383-412, with
# PIL-backed stand-ins for the absent torchvision transforms) and stored the crops it handed to the model.  The frames
# came from default_rng(0) in a fixed order; replaying the draws gives the same pixels.
def _reference_frames():
    rng = np.random.default_rng(0)
    frame1 = rng.integers(0, 256, (720, 1280, 3), dtype=np.uint8)
    rng.integers(0, 256, (128, 96, 3), dtype=np.uint8)
    rng.integers(0, 256, (5, 128, 96, 3), dtype=np.uint8)
    frames3 = rng.integers(0, 256, (3, 480, 640, 3), dtype=np.uint8)
    return frame1, frames3


DETS_SINGLE = np.asarray([[100.2, 50.7, 400.4, 650.1], [600.0, 200.0, 1100.0, 500.0], [5.0, 300.0, 250.0, 700.0]], np.float32)
DETS_BATCH = {0: np.asarray([[50., 40., 200., 400.], [300., 100., 620., 300.]], np.float32),
              2: np.asarray([[10., 10., 630., 470.]], np.float32)}


def test_oracle_equals_reference_predict_crops():
    frame1, frames3 = _reference_frames()
    g = golden("cfg1_w32_256x192_predict_multi")
    images, boxes = P.prepath(frame1, DETS_SINGLE, 256, 192)
    np.testing.assert_array_equal(boxes, g["boxes"])
    np.testing.assert_array_equal(images, g["crops"])
    g = golden("w32_128x96_predict_batch_multi")
    parts = [P.prepath_clamped(frames3[d], DETS_BATCH[d], 128, 96) for d in (0, 2)]
    np.testing.assert_array_equal(np.concatenate([p[1] for p in parts]), g["boxes"])
    np.testing.assert_array_equal(np.concatenate([p[0] for p in parts]), g["crops"])


@pytest.mark.gpu
def test_gpu_equals_reference_predict_crops(pkg):
    frame1, frames3 = _reference_frames()
    g = golden("cfg1_w32_256x192_predict_multi")
    net = pkg.NativeHRNet(32, 17, (256, 192), "fp32", max_batch=4, device=0).load_state_dict(state_dict_np(32, 0))
    images, boxes, boxes_dev = net.preprocess_frame(frame1, DETS_SINGLE)
    np.testing.assert_array_equal(boxes, g["boxes"])
    np.testing.assert_array_equal(images.cpu().numpy(), g["crops"])
    # the whole of predict() after the detector: frame -> joints, against the reference's own output
    bx, pts, hm = net.predict_frame(frame1, DETS_SINGLE, return_heatmaps=True)
    np.testing.assert_allclose(hm.cpu().numpy(), g["heatmaps"], rtol=0, atol=2e-4)
    np.testing.assert_array_equal(pts.cpu().numpy()[..., :2], g["pts"].reshape(3, 17, 3)[..., :2])
    net.close()
    g = golden("w32_128x96_predict_batch_multi")
    net = pkg.NativeHRNet(32, 17, (128, 96), "fp32", max_batch=4, device=0)
    outs = [net.preprocess_frame(frames3[d], DETS_BATCH[d], "clamp") for d in (0, 2)]
    np.testing.assert_array_equal(np.concatenate([o[1] for o in outs]), g["boxes"])
    np.testing.assert_array_equal(np.concatenate([o[0].cpu().numpy() for o in outs]), g["crops"])
    net.close()
