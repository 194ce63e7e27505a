"""Greedy IoU NMS (misc/nms/): the restated python NMS against the reference function's own
outputs (fixture), then ``hrn_nms`` / ``gpu_nms`` against both."""
import numpy as np
import pytest

from conftest import golden, load_pkg
from oracle import nms_oracle


def _cases():
    g = golden("nms_cases")
    return [(g["dets%d" % i], float(g["thresh%d" % i]), g["keep%d" % i]) for i in range(int(g["ncases"]))]


def test_oracle_matches_reference_nms():
    for dets, thr, keep in _cases():
        assert nms_oracle.nms(dets, thr) == keep.tolist()
    assert nms_oracle.nms(np.zeros((0, 5), np.float32), 0.5) == []


def _boxes(n, seed):
    rng = np.random.default_rng(seed)
    centres = rng.uniform([40, 40], [600, 440], size=(max(1, n // 8), 2))
    c = centres[rng.integers(0, len(centres), n)] + rng.normal(0, 12, (n, 2))
    wh = rng.uniform(20, 160, (n, 2))
    return np.concatenate([c - wh / 2, c + wh / 2, rng.permutation(n)[:, None] / n + 0.001], 1).astype(np.float32)


@pytest.mark.gpu
def test_gpu_nms_matches_reference_and_oracle():
    nms_mod = load_pkg("nms")
    for dets, thr, keep in _cases():
        assert [int(i) for i in nms_mod.gpu_nms(dets, thr)] == keep.tolist()
    for n, thr, seed in [(2, 0.5, 10), (63, 0.4, 11), (64, 0.6, 12), (129, 0.5, 13), (2000, 0.5, 14), (4096, 0.35, 15)]:
        dets = _boxes(n, seed)
        extra = np.concatenate([dets, np.zeros((n, 2), np.float32)], 1)          # boxes_dim 7, like detector rows
        ref = nms_oracle.nms(dets, thr)
        assert [int(i) for i in nms_mod.gpu_nms(dets, thr)] == ref
        assert [int(i) for i in nms_mod.gpu_nms(extra, thr)] == ref
    assert nms_mod.gpu_nms(np.zeros((0, 5), np.float32), 0.5) == []
    with pytest.raises(ValueError):
        nms_mod.gpu_nms(np.zeros((3, 4), np.float32), 0.5)
