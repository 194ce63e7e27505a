"""The workspace invariant (ctx_plan.inc: new_tensor): the pad column / pad row of every image and the guard rows of
every activation buffer hold zeros at all times -- every 3x3 convolution's zero padding IS those positions, and the stride-2
slab kernel never rewrites them.  `hrn_debug_pad_violations` counts what is not zero there; after every path of the engine has
run -- both models, both dtypes, ragged micro-batches, small and large calls (128- / 512-pixel tiles, fused and plain BasicBlocks,
slab kernel and its generic fallback, the work-queue form), flip-TTA -- it must say 0."""
import numpy as np
import pytest
import torch

from conftest import load_pkg, state_dict_np

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("c,res,dtype,mb,calls,env", [
    (48, (384, 288), "bf16", 256, (256, 3, 100), {}),
    (48, (384, 288), "bf16", 147, (147, 19), {"HRN_SMALL_TILES": "0"}),   # nearly empty last compact tiles at n == max_batch
    (48, (128, 96), "bf16", 8, (8, 3, 1), {"HRN_S2_MIN_TILES": "1", "HRN_BBF_MIN_TILES": "1"}),
    (32, (256, 192), "fp32", 16, (16, 5), {}),
    (32, (256, 192), "bf16", 64, (64, 7), {}),
])
def test_pad_positions_stay_zero_hrnet(monkeypatch, c, res, dtype, mb, calls, env):
    assert torch.cuda.is_available(), "GPU tests need a GPU: the HIP path has no CPU fallback"
    pkg = load_pkg()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    net = pkg.NativeHRNet(c, 17, res, dtype, max_batch=mb, device=0).load_state_dict(state_dict_np(c))
    assert net.pad_violations() == 0                                   # freshly allocated: all zeros
    for n in calls:
        crops = torch.from_numpy(pkg.synth_crops(n, res[0], res[1], seed=70 + n)).cuda()
        pts = net.predict_crops(crops, pkg.synth_boxes(n, seed=n))
        assert bool(torch.isfinite(pts).all())
        assert net.pad_violations() == 0, "a kernel left something in a pad / guard position (call of %d crops)" % n
    crops = torch.from_numpy(pkg.synth_crops(min(mb, 4), res[0], res[1], seed=9)).cuda()
    net.predict_flip_tta(crops, [[1, 2], [3, 4]])
    assert net.pad_violations() == 0
    net.close()


def test_pad_positions_stay_zero_poseresnet():
    pkg = load_pkg()
    sd = pkg.synth_state_dict(50, 17, 0, model="PoseResNet")
    for dtype in ("bf16", "fp32"):
        net = pkg.NativeHRNet(50, 17, (256, 192), dtype, max_batch=8, device=0, model_name="PoseResNet").load_state_dict(sd)
        for n in (8, 3):
            net.predict_crops(torch.from_numpy(pkg.synth_crops(n, 256, 192, seed=n)).cuda(), pkg.synth_boxes(n, seed=n))
            assert net.pad_violations() == 0, (dtype, n)
        net.close()
