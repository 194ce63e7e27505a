"""``SimpleHRNet`` facade: the reference's constructor + ``predict()`` contract (SimpleHRNet.py:21-172, 174-496).

The GPU tests read like reference usage -- build ``SimpleHRNet(c, joints, checkpoint, ...)``, call ``predict(frame)`` --
and compare with what the unmodified reference's ``predict()`` returned for the same frames, checkpoint and detector
table (tests/golden/make_golden.py ``predict_cases``).  Tolerances: boxes and crops bit-exact, heat-maps 2e-4 absolute
(fp32 summation order), joint coordinates exact (well inside north_star's +-0.5 px)."""
import os

import numpy as np
import pytest
import torch

from conftest import golden, load_pkg, state_dict_np
from test_prepath import DETS_BATCH, DETS_SINGLE


def _frames():
    rng = np.random.default_rng(0)   # the draw order of make_golden.predict_cases
    return (rng.integers(0, 256, (720, 1280, 3), dtype=np.uint8), rng.integers(0, 256, (128, 96, 3), dtype=np.uint8),
            rng.integers(0, 256, (5, 128, 96, 3), dtype=np.uint8), rng.integers(0, 256, (3, 480, 640, 3), dtype=np.uint8))


class TableDetector:
    """the YOLOv3 stand-in of make_golden.py: per image a (P,7) tensor or None (models_/detectors/YOLOv3.py:79-141)"""

    def __init__(self, table):
        self.table = {k: (None if v is None else torch.cat([torch.as_tensor(v), torch.tensor([[.9, .9, 0.]] * len(v))], 1))
                      for k, v in table.items()}

    def predict_single(self, image, color_mode="BGR"):
        return self.predict(np.expand_dims(image, 0))[0]

    def predict(self, images, color_mode="BGR"):
        return [self.table.get(i) for i in range(len(images))]


def test_constructor_errors_without_gpu():
    S = load_pkg().SimpleHRNet
    with pytest.raises(ValueError, match="Wrong model name."):
        S(32, 17, {}, model_name="resnext")
    with pytest.raises(ValueError):
        S(32, 17, {}, device=torch.device("cpu"))
    with pytest.raises(ValueError, match="detector"):
        S(32, 17, {}, multiperson=True)
    with pytest.raises(ValueError):
        S(32, 17, {}, enable_tensorrt=True)


def test_device_strings_of_the_reference_map_to_one_gpu_per_process(monkeypatch):
    from importlib import import_module
    resolve = import_module("simple-hrnet_amd.simple_hrnet").resolve_device
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    assert resolve("cuda:3") == torch.device("cuda", 3) and resolve(torch.device("cuda:2")) == torch.device("cuda", 2)
    assert resolve("cuda") == torch.device("cuda", 0) and resolve(None) == torch.device("cuda", 0)
    assert resolve("cuda:1,2") == torch.device("cuda", 1)
    monkeypatch.setenv("LOCAL_RANK", "1")                       # as set by torch.distributed.run
    assert resolve("cuda") == torch.device("cuda", 1) and resolve("cuda:4,6") == torch.device("cuda", 6)
    assert resolve("cuda:5") == torch.device("cuda", 5)         # an explicit single GPU is taken literally
    for bad in ("cuda:x", "cuda:-1", "cuda:"):
        with pytest.raises(ValueError, match="Wrong device name."):
            resolve(bad)
    for cpu in ("cpu", torch.device("cpu")):
        with pytest.raises(ValueError, match="no CPU path"):
            resolve(cpu)


class _FakeEngine:
    """stands in for NativeHRNet on a box without a GPU: joints encode (running crop number, joint) so that the result
    assembly of SimpleHRNet.predict -- what is tested here -- can be followed crop by crop"""

    def __init__(self, c, nof_joints, resolution, dtype, max_batch, device, model_name):
        self.j, self.res, self.count = nof_joints, resolution, 0

    def load_state_dict(self, sd):
        return self

    def preprocess_frame(self, frame, dets, variant="pad"):
        assert variant in ("pad", "clamp") and frame.ndim == 3
        boxes = np.rint(np.asarray(dets, np.float32)[:, :4]).astype(np.int32)
        return torch.zeros((len(dets), 3) + tuple(self.res)), boxes, torch.from_numpy(boxes)

    def predict_crops(self, images, boxes, return_heatmaps=False):
        n = images.shape[0]
        pts = torch.zeros((n, self.j, 3))
        pts[:, :, 0] = torch.arange(self.count, self.count + n, dtype=torch.float32)[:, None]
        pts[:, :, 1] = torch.arange(self.j, dtype=torch.float32)[None, :]
        self.count += n
        hm = torch.zeros((n, self.j, self.res[0] // 4, self.res[1] // 4))
        return (hm, pts) if return_heatmaps else pts

    def predict_frame(self, frame, dets, return_heatmaps=False, variant="pad"):
        images, boxes, boxes_dev = self.preprocess_frame(frame, dets, variant)
        out = self.predict_crops(images, boxes_dev, return_heatmaps)
        return (boxes, out[1], out[0]) if return_heatmaps else (boxes, out)


def test_result_assembly_follows_the_reference_without_a_gpu(monkeypatch):
    """SimpleHRNet.py:333-343 / 445-496 on a fake engine: list order, bare-array rule, per-image re-split, empty shapes."""
    from importlib import import_module
    mod = import_module("simple-hrnet_amd.simple_hrnet")
    monkeypatch.setattr(mod, "NativeHRNet", _FakeEngine)
    frames = np.zeros((4, 60, 80, 3), np.uint8)
    table = {0: np.asarray([[1, 2, 30, 50], [5, 5, 20, 40]], np.float32), 1: None, 2: np.zeros((0, 4), np.float32),
             3: np.asarray([[0, 0, 79, 59]], np.float32)}
    m = mod.SimpleHRNet(32, 17, {}, resolution=(64, 48), multiperson=True, return_heatmaps=True, return_bounding_boxes=True,
                        detector=TableDetector(table))
    hm, boxes, pts = m.predict(frames)
    assert [len(p) for p in pts] == [2, 0, 0, 1] and all(p.shape[1:] == (17, 3) and p.dtype == np.float32 for p in pts)
    assert [h.shape for h in hm] == [(2, 17, 16, 12), (0, 17, 16, 12), (0, 17, 16, 12), (1, 17, 16, 12)]
    assert [b.shape for b in boxes] == [(2, 4), (0, 4), (0, 4), (1, 4)]
    assert pts[0][:, 0, 0].tolist() == [0.0, 1.0] and pts[3][:, 0, 0].tolist() == [2.0]     # crops kept in image order
    np.testing.assert_array_equal(boxes[3], [[0, 0, 79, 59]])
    m.return_heatmaps = False
    boxes, pts = m.predict(frames)
    assert isinstance(boxes, list) and len(pts) == 4
    m.return_bounding_boxes = False
    pts = m.predict(frames)
    assert isinstance(pts, list) and pts[0].shape == (2, 17, 3)
    # one image
    m.return_heatmaps = m.return_bounding_boxes = True
    hm, boxes, pts = m.predict(frames[0])
    assert hm.shape == (2, 17, 16, 12) and boxes.shape == (2, 4) and boxes.dtype == np.int32 and pts.shape == (2, 17, 3)
    m.detector = TableDetector({0: np.zeros((0, 4), np.float32)})
    hm, boxes, pts = m.predict(frames[0])
    assert pts.shape == (0, 0, 3) and boxes.shape == (0, 4) and hm.shape == (0, 17, 16, 12)
    with pytest.raises(ValueError, match="Wrong image format."):
        m.predict(frames[0, 0])


@pytest.mark.gpu
def test_predict_single_image_multiperson_equals_reference(tmp_path):
    pkg = load_pkg()
    ck = os.path.join(tmp_path, "w32.pth")
    torch.save(pkg.synth.to_torch_state_dict(state_dict_np(32, 0)), ck)
    g = golden("cfg1_w32_256x192_predict_multi")
    model = pkg.SimpleHRNet(32, 17, ck, resolution=(256, 192), multiperson=True, return_heatmaps=True,
                            return_bounding_boxes=True, device=torch.device("cuda:0"),
                            detector=TableDetector({0: DETS_SINGLE}))
    hm, boxes, pts = model.predict(_frames()[0])
    assert hm.dtype == np.float32 and pts.dtype == np.float32 and boxes.dtype == np.int32
    np.testing.assert_array_equal(boxes, g["boxes"])
    np.testing.assert_allclose(hm, g["heatmaps"], rtol=0, atol=2e-4)
    np.testing.assert_array_equal(pts[..., :2], g["pts"][..., :2])
    np.testing.assert_allclose(pts[..., 2], g["pts"][..., 2], rtol=0, atol=2e-4)
    # return-structure rules of :333-343: bare joints when nothing else is asked for
    model.return_heatmaps = model.return_bounding_boxes = False
    only = model.predict(_frames()[0])
    assert isinstance(only, np.ndarray) and only.shape == (3, 17, 3)
    # nobody in the frame (:331)
    model.detector = TableDetector({0: None})
    assert model.predict(_frames()[0]).shape == (0, 0, 3)
    with pytest.raises(ValueError, match="Wrong image format."):
        model.predict(np.zeros((4, 4), np.uint8))


@pytest.mark.gpu
def test_predict_single_person_paths_equal_reference():
    pkg = load_pkg()
    _, frame, frames5, _ = _frames()
    g = golden("w32_128x96_predict_single")
    model = pkg.SimpleHRNet(32, 17, state_dict_np(32, 0), resolution=(128, 96), multiperson=False, return_heatmaps=True,
                            return_bounding_boxes=True, device="cuda:0")
    np.testing.assert_array_equal(model._normalise(frame).cpu().numpy(), g["crops"])
    hm, boxes, pts = model.predict(frame)
    assert boxes.dtype == np.float32
    np.testing.assert_array_equal(boxes, g["boxes"])
    np.testing.assert_allclose(hm, g["heatmaps"], rtol=0, atol=2e-4)
    np.testing.assert_array_equal(pts[..., :2], g["pts"][..., :2])
    # frames of another size are resized on the GPU (tests/test_resize.py); whole-frame box in the frame's own pixels (:223)
    assert model.predict(np.zeros((64, 80, 3), np.uint8))[1].tolist() == [[0, 0, 80, 64]]
    # stack of 5 through an engine sized for 2 -> the chunk loop of :423-429
    g = golden("w48_128x96_predict_batch5")
    model = pkg.SimpleHRNet(48, 17, state_dict_np(48, 0), resolution=(128, 96), multiperson=False, return_heatmaps=True,
                            return_bounding_boxes=True, max_batch_size=2, device="cuda:0")
    hm, boxes, pts = model.predict(frames5)
    assert pts.shape == g["pts"].shape == (5, 1, 17, 3)
    np.testing.assert_array_equal(boxes, g["boxes"])
    np.testing.assert_allclose(hm, g["heatmaps"], rtol=0, atol=2e-4)
    np.testing.assert_array_equal(pts[..., :2], g["pts"][..., :2])


@pytest.mark.gpu
def test_predict_stack_multiperson_equals_reference():
    pkg = load_pkg()
    frames3 = _frames()[3]
    g = golden("w32_128x96_predict_batch_multi")
    model = pkg.SimpleHRNet(32, 17, state_dict_np(32, 0), resolution=(128, 96), multiperson=True, return_heatmaps=True,
                            return_bounding_boxes=True, device="cuda:0",
                            detector=TableDetector({0: DETS_BATCH[0], 1: None, 2: DETS_BATCH[2]}))
    hm, boxes, pts = model.predict(frames3)
    assert [len(p) for p in pts] == list(g["counts"]) == [2, 0, 1]
    assert pts[1].shape == (0, 17, 3) and hm[1].shape == (0, 17, 32, 24) and boxes[1].shape == (0, 4)
    np.testing.assert_array_equal(np.concatenate(boxes, 0), g["boxes"])
    np.testing.assert_allclose(np.concatenate(hm, 0), g["heatmaps"], rtol=0, atol=2e-4)
    np.testing.assert_array_equal(np.concatenate(pts, 0)[..., :2], g["pts"][..., :2])
    # nobody in any image (:477-484)
    model.detector = TableDetector({})
    hm, boxes, pts = model.predict(frames3)
    assert len(pts) == 3 and all(p.shape == (0, 17, 3) for p in pts) and boxes.shape == (0,) and boxes.dtype == np.int32
