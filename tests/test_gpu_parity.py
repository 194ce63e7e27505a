"""Parity of the HIP path (through the C ABI) with the reference's outputs -- needs a real MI355X.

fp32 engine: the reference's own outputs (tests/golden/*.npz) and the oracle on further seeded inputs;
bar = identical arg-max for every (crop, joint)  =>  identical coordinates (<< 0.5 px), heat-maps within
2e-4 abs (heat-map sigma ~2.5-5.7; fp32 summation-order noise measured ~1e-5).
bf16 engine: heat-map error bound + arg-max agreement wherever the oracle's top-1/top-2 gap exceeds 4x the
measured heat-map error + pixel-deviation report.
"""
import numpy as np
import pytest
import torch

from conftest import golden, load_pkg, state_dict_np

pytestmark = pytest.mark.gpu

HM_ATOL_F32 = 2e-4


@pytest.fixture(scope="module")
def pkg():
    p = load_pkg()
    assert torch.cuda.is_available(), "GPU tests need a GPU: the HIP path has no CPU fallback"
    return p


def _oracle():
    from oracle import hrnet_torch_oracle as T
    return T


def _crops(g, pkg):
    return g["crops"] if "crops" in g else pkg.synth_crops(int(g["n"]), int(g["h"]), int(g["w"]))


def _engine(pkg, c, h, w, dtype, max_batch, seed=0):
    net = pkg.NativeHRNet(c, 17, (h, w), dtype, max_batch=max_batch, device=0)
    net.load_state_dict(state_dict_np(c, seed))
    return net


GOLDEN = ["w32_64x64_n2", "w48_64x64_n2", "w32_256x192_n2", "w48_384x288_n1", "cfg1_w32_256x192_predict_multi",
          "w32_128x96_predict_single", "w48_128x96_predict_batch5", "w32_128x96_predict_batch_multi"]


@pytest.mark.parametrize("name", GOLDEN)
def test_fp32_matches_reference_outputs(pkg, name):
    g = golden(name)
    c, h, w = int(g["c"]), int(g["h"]), int(g["w"])
    crops = torch.from_numpy(_crops(g, pkg)).cuda()
    net = _engine(pkg, c, h, w, "fp32", max_batch=2, seed=int(g["weight_seed"]))  # max_batch 2 -> chunk loop
    hm, pts = net.predict_crops(crops, g["boxes"], return_heatmaps=True)
    hm, pts = hm.cpu().numpy(), pts.cpu().numpy()
    ref_hm, ref_pts = g["heatmaps"], g["pts"].reshape(pts.shape)
    assert np.isfinite(hm).all()
    np.testing.assert_allclose(hm, ref_hm, rtol=0, atol=HM_ATOL_F32)
    # identical arg-max -> coordinates bit-identical to the reference's float64->float32 arithmetic
    np.testing.assert_array_equal(pts[..., :2], ref_pts[..., :2])
    np.testing.assert_allclose(pts[..., 2], ref_pts[..., 2], rtol=0, atol=HM_ATOL_F32)
    # level-1 seam: self.model(images) returns the same heat-maps
    hm2 = net(crops).cpu().numpy()
    np.testing.assert_array_equal(hm2, hm)
    net.close()


@pytest.mark.parametrize("c,h,w,n,mb", [(32, 64, 96, 5, 3), (48, 96, 64, 4, 4), (32, 128, 128, 1, 8)])
def test_fp32_vs_oracle_seeded(pkg, c, h, w, n, mb):
    T = _oracle()
    sd = pkg.synth.to_torch_state_dict(state_dict_np(c, 7))
    crops = pkg.synth_crops(n, h, w, seed=11)
    boxes = pkg.synth_boxes(n, seed=5)
    ref_hm, ref_pts = T.predict_crops(sd, torch.from_numpy(crops), boxes)
    net = _engine(pkg, c, h, w, "fp32", max_batch=mb, seed=7)
    hm, pts = net.predict_crops(torch.from_numpy(crops).cuda(), boxes, return_heatmaps=True)
    np.testing.assert_allclose(hm.cpu().numpy(), ref_hm, rtol=0, atol=HM_ATOL_F32)
    np.testing.assert_array_equal(pts.cpu().numpy()[..., :2], ref_pts[..., :2])
    # float32 boxes (single-person path, SimpleHRNet.py:223)
    bf = boxes.astype(np.float32) + np.float32(0.37)
    pts_f = net.predict_crops(torch.from_numpy(crops).cuda(), bf).cpu().numpy()
    np.testing.assert_array_equal(pts_f[..., :2], T.decode_heatmaps(ref_hm, bf)[..., :2])
    net.close()


def test_empty_batch_and_errors(pkg):
    net = _engine(pkg, 32, 64, 64, "fp32", 2)
    out = net(torch.empty((0, 3, 64, 64), device="cuda"))
    assert tuple(out.shape) == (0, 17, 16, 16)
    pts = net.predict_crops(torch.empty((0, 3, 64, 64), device="cuda"), np.zeros((0, 4), np.int32))
    assert tuple(pts.shape) == (0, 17, 3)
    with pytest.raises(ValueError):
        net(torch.zeros((1, 3, 64, 32), device="cuda"))
    fresh = pkg.NativeHRNet(32, 17, (64, 64), "fp32", max_batch=2, device=0)
    with pytest.raises(RuntimeError, match="weights not loaded"):
        fresh(torch.zeros((1, 3, 64, 64), device="cuda"))
    fresh.close()
    net.close()


def _bf16_report(hm, ref_hm):
    err = np.abs(hm - ref_hm)
    flat, rflat = hm.reshape(*hm.shape[:2], -1), ref_hm.reshape(*hm.shape[:2], -1)
    am, ram = flat.argmax(-1), rflat.argmax(-1)
    top2 = np.sort(rflat, -1)[..., -2:]
    gap = top2[..., 1] - top2[..., 0]
    return err.max(), am, ram, gap


@pytest.mark.parametrize("name", ["w32_64x64_n2", "w48_64x64_n2", "w32_256x192_n2", "w48_384x288_n1",
                                  "cfg1_w32_256x192_predict_multi"])
def test_bf16_bounded_error_and_argmax(pkg, name):
    g = golden(name)
    c, h, w = int(g["c"]), int(g["h"]), int(g["w"])
    crops = torch.from_numpy(_crops(g, pkg)).cuda()
    net = _engine(pkg, c, h, w, "bf16", max_batch=4, seed=int(g["weight_seed"]))
    hm, pts = net.predict_crops(crops, g["boxes"], return_heatmaps=True)
    hm, pts = hm.cpu().numpy(), pts.cpu().numpy()
    ref_hm = g["heatmaps"]
    assert np.isfinite(hm).all()
    err, am, ram, gap = _bf16_report(hm, ref_hm)
    sigma = ref_hm.std()
    print("\n[bf16 %s] max|dH|=%.4f (sigma %.2f, rel %.4f)  argmax agree %d/%d  min gap %.4f" %
          (name, err, sigma, err / sigma, (am == ram).sum(), am.size, gap.min()))
    # bf16 activations carried through ~110 residual layers: bound relative to the heat-map scale
    assert err < 0.05 * sigma + 0.05
    # wherever the reference's decision margin exceeds 4x the error, the arg-max must agree
    decided = gap > 4 * err
    assert (am == ram)[decided].all()
    # pixel deviation of the joints (heat-map cell = 4 px of the crop): report + sanity bound on the median
    hw = ref_hm.shape[-1]
    dy = np.abs(am // hw - ram // hw)
    dx = np.abs(am % hw - ram % hw)
    print("   cell deviation histogram (max(dy,dx)):", np.bincount(np.maximum(dy, dx).ravel())[:8])
    assert np.median(np.maximum(dy, dx)) == 0
    net.close()


def test_full_size_properties_bf16(pkg):
    """BASELINE config sizes (W48 384x288) where the CPU oracle would take minutes: size-independent
    properties -- determinism, independence of a crop from its batch position / micro-batch size."""
    c, h, w, n = 48, 384, 288, 24
    crops = torch.from_numpy(pkg.synth_crops(n, h, w, seed=21)).cuda()
    boxes = pkg.synth_boxes(n, seed=9)
    a = _engine(pkg, c, h, w, "bf16", max_batch=16)
    hm1, p1 = a.predict_crops(crops, boxes, return_heatmaps=True)
    hm2, p2 = a.predict_crops(crops, boxes, return_heatmaps=True)
    assert torch.equal(hm1, hm2) and torch.equal(p1, p2)                      # deterministic
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(0))
    hm3, p3 = a.predict_crops(crops[perm.cuda()], boxes[perm.numpy()], return_heatmaps=True)
    assert torch.equal(hm3, hm1[perm.cuda()]) and torch.equal(p3, p1[perm.cuda()])   # position independent
    a.close()
    b = _engine(pkg, c, h, w, "bf16", max_batch=5)                            # different chunking (5,5,5,5,4)
    hm4, p4 = b.predict_crops(crops, boxes, return_heatmaps=True)
    assert torch.equal(hm4, hm1) and torch.equal(p4, p1)
    # decode consistency: pts recomputed from the returned heat-maps by the oracle's decode
    T = _oracle()
    np.testing.assert_array_equal(T.decode_heatmaps(hm1.cpu().numpy(), boxes), p1.cpu().numpy())
    b.close()


def test_weight_blob_broadcast_path_single_gpu(pkg):
    """The multi-GPU weight distribution minus RCCL: the packed blob of a loaded engine is exposed as a
    zero-copy uint8 CUDA tensor, copied into a second (unloaded) engine's blob and adopted -- what
    ShardedHRNet.load_and_broadcast does with dist.broadcast -- and both engines then agree bit for bit."""
    c, h, w, n = 32, 64, 64, 3
    src = _engine(pkg, c, h, w, "bf16", 4)
    dst = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=4, device=0)
    a, b = src.weight_blob_tensor(), dst.weight_blob_tensor()
    assert a.dtype == torch.uint8 and a.is_cuda and a.numel() == src.weight_blob_bytes() == b.numel()
    assert a.data_ptr() != b.data_ptr()
    b.copy_(a)
    torch.cuda.synchronize()
    dst.adopt_weights()
    crops = torch.from_numpy(pkg.synth_crops(n, h, w, seed=4)).cuda()
    boxes = pkg.synth_boxes(n, seed=4)
    p1, p2 = src.predict_crops(crops, boxes), dst.predict_crops(crops, boxes)
    assert torch.equal(p1, p2)
    # and through the sharding front-end with a single process (world size 1)
    sh = load_pkg("dist").ShardedHRNet(dst, None)
    assert torch.equal(sh.predict_crops_sharded(crops, boxes), p1)
    assert torch.equal(sh.predict_crops_local_then_gather(crops, boxes), p1)
    src.close()
    dst.close()


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_w32_config2_shape_batch64(pkg, dtype):
    """BASELINE configs[1]: HRNet-W32 256x192, batch 64, one GPU.  The oracle at this size takes ~3 s on CPU,
    so it is checked on the first 4 crops; the rest by batch-position invariance."""
    T = _oracle()
    c, h, w, n = 32, 256, 192, 64
    crops_np = pkg.synth_crops(n, h, w, seed=8)
    boxes = pkg.synth_boxes(n, seed=8)
    net = _engine(pkg, c, h, w, dtype, max_batch=64)
    hm, pts = net.predict_crops(torch.from_numpy(crops_np).cuda(), boxes, return_heatmaps=True)
    sd = pkg.synth.to_torch_state_dict(state_dict_np(c))
    ref_hm, ref_pts = T.predict_crops(sd, torch.from_numpy(crops_np[:4]), boxes[:4])
    if dtype == "fp32":
        np.testing.assert_allclose(hm[:4].cpu().numpy(), ref_hm, rtol=0, atol=HM_ATOL_F32)
        np.testing.assert_array_equal(pts[:4].cpu().numpy()[..., :2], ref_pts[..., :2])
    else:
        assert np.abs(hm[:4].cpu().numpy() - ref_hm).max() < 0.05 * ref_hm.std() + 0.05
    hm_b, pts_b = net.predict_crops(torch.from_numpy(crops_np[::-1].copy()).cuda(), boxes[::-1].copy(), return_heatmaps=True)
    assert torch.equal(hm_b.flip(0), hm) and torch.equal(pts_b.flip(0), pts)
    net.close()


def test_ragged_micro_batches_and_blockmap_rebuild(pkg):
    """n not a multiple of max_batch: the tail pass runs with a different row count, which re-derives the block
    maps of the grouped conv launches; alternate sizes back and forth and compare every crop with its
    stand-alone result (bit-exact: a crop's arithmetic does not depend on its neighbours)."""
    c, h, w = 48, 128, 96
    net = _engine(pkg, c, h, w, "bf16", max_batch=8)
    crops = torch.from_numpy(pkg.synth_crops(21, h, w, seed=31)).cuda()
    boxes = pkg.synth_boxes(21, seed=31)
    ref_hm, ref_pts = net.predict_crops(crops[:8], boxes[:8], return_heatmaps=True)
    for n in (21, 1, 9, 8, 3, 21):
        hm, pts = net.predict_crops(crops[:n], boxes[:n], return_heatmaps=True)
        k = min(n, 8)
        assert torch.equal(hm[:k], ref_hm[:k]) and torch.equal(pts[:k], ref_pts[:k]), n
        if n == 21:  # crops 16..20 went through the tail pass (nb = 5): same bits as a batch that starts with them
            hm2, pts2 = net.predict_crops(crops[16:21], boxes[16:21], return_heatmaps=True)
            assert torch.equal(hm[16:], hm2) and torch.equal(pts[16:], pts2)
    net.close()


def test_repeatability_full_size(pkg):
    """race screen for the hand-scheduled kernels (inline-asm LDS reads, counted waits, LDS-DMA pipeline):
    300 W48 384x288 crops, micro-batch 256, three runs must agree bit for bit."""
    c, h, w, n = 48, 384, 288, 300
    net = _engine(pkg, c, h, w, "bf16", max_batch=256)
    g = torch.Generator(device="cuda").manual_seed(5)
    crops = torch.randn((n, 3, h, w), generator=g, device="cuda")
    boxes = pkg.synth_boxes(n, seed=5)
    hm0, p0 = net.predict_crops(crops, boxes, return_heatmaps=True)
    for _ in range(2):
        hm, p = net.predict_crops(crops, boxes, return_heatmaps=True)
        assert torch.equal(hm, hm0) and torch.equal(p, p0)
    # the first 44 crops of the tail pass (nb = 44) equal the same crops run on their own
    hm_t, p_t = net.predict_crops(crops[256:], boxes[256:], return_heatmaps=True)
    assert torch.equal(hm_t, hm0[256:]) and torch.equal(p_t, p0[256:])
    assert bool(torch.isfinite(hm0).all())
    net.close()


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_scheduling_variants_are_bit_identical(pkg, dtype, monkeypatch):
    """Launch grouping, block order and walk direction only reschedule work: every variant must give the same bits.
    (The switches are read when the handle is created.)"""
    c, h, w, n = 48, 128, 96, 5
    crops = torch.from_numpy(pkg.synth_crops(n, h, w, seed=21)).cuda()
    boxes = pkg.synth_boxes(n, seed=22)

    def run(env):
        for k in ("HRN_DISABLE_DGROUP", "HRN_DISABLE_CHAIN", "HRN_DISABLE_CHAIN_DS", "HRN_DISABLE_CHAIN3", "HRN_BBF", "HRN_BBF_MIN_TILES", "HRN_DIRECT_WLDS", "HRN_DIRECT_XLDS", "HRN_SMALL_KEEP", "HRN_SMALL_TILES", "HRN_DGROUP_NR", "HRN_ALTERNATE", "HRN_DISABLE_GROUP", "HRN_DISABLE_FGROUP", "HRN_BLOCK_ORDER",
                  "HRN_LONG_FACTOR"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        net = _engine(pkg, c, h, w, dtype, max_batch=4, seed=3)   # 5 crops, micro-batch 4: ragged second pass
        hm, pts = net.predict_crops(crops, boxes, return_heatmaps=True)
        out = hm.cpu().numpy(), pts.cpu().numpy()
        net.close()
        return out

    base = run({})
    for env in ({"HRN_DISABLE_DGROUP": "1"}, {"HRN_DISABLE_CHAIN": "1"}, {"HRN_DISABLE_CHAIN_DS": "1"}, {"HRN_DISABLE_CHAIN3": "1"}, {"HRN_BBF": "0"}, {"HRN_BBF_MIN_TILES": "1"}, {"HRN_DIRECT_WLDS": "0"}, {"HRN_DIRECT_XLDS": "0"}, {"HRN_SMALL_KEEP": "0"}, {"HRN_SMALL_TILES": "0"}, {"HRN_DGROUP_NR": "0"}, {"HRN_ALTERNATE": "0"}, {"HRN_DISABLE_GROUP": "1"}, {"HRN_DISABLE_FGROUP": "1"},
                {"HRN_BLOCK_ORDER": "0", "HRN_LONG_FACTOR": "1"}):
        hm, pts = run(env)
        np.testing.assert_array_equal(hm, base[0], err_msg=str(env))
        np.testing.assert_array_equal(pts, base[1], err_msg=str(env))


@pytest.mark.parametrize("h,w,n,mb", [(384, 288, 3, 4), (384, 288, 7, 3), (256, 192, 5, 8), (64, 64, 9, 16)])
def test_fused_basicblock_is_bit_identical(pkg, monkeypatch, h, w, n, mb):
    """The fused BasicBlock pass of the 48-channel branch (conv3x3_lds.hip: bbf_run) against the two separate launches:
    same bits, at the row pitches that deal conv1's fragments 6/5, 5/4 and 4 per wave, with ragged last micro-batches."""
    crops = torch.from_numpy(pkg.synth_crops(n, h, w, seed=41)).cuda()
    boxes = pkg.synth_boxes(n, seed=42)
    outs = []
    monkeypatch.setenv("HRN_BBF_MIN_TILES", "1")   # (by default only calls with >= 1100 tiles of 512 pixels take the fused pass)
    for on in (True, False):
        monkeypatch.delenv("HRN_BBF", raising=False)
        if not on:
            monkeypatch.setenv("HRN_BBF", "0")
        net = _engine(pkg, 48, h, w, "bf16", max_batch=mb, seed=3)
        hm, pts = net.predict_crops(crops, boxes, return_heatmaps=True)
        bb0 = [i for i in net.conv_infos() if b".branches.0." in bytes(i.name)]
        outs.append((hm.cpu().numpy(), pts.cpu().numpy(), sum(i.algo == 2 for i in net.conv_infos()), len(bb0)))
        net.close()
    assert np.isfinite(outs[0][0]).all()
    assert outs[0][3] > 0 and outs[0][2] == outs[0][3] and outs[1][2] == 0   # all 32 BasicBlocks of the 48-channel branch went through the fused pass
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("model,c,h,w,n,mb", [("HRNet", 48, 384, 288, 5, 4), ("HRNet", 32, 256, 192, 33, 16), ("HRNet", 48, 64, 64, 9, 16),
                                              ("HRNet", 48, 384, 288, 130, 130), ("PoseResNet", 50, 256, 192, 6, 4)])
def test_bottleneck_3x3_inside_the_chain_kernel_is_bit_identical(pkg, monkeypatch, model, c, h, w, n, mb):
    """Round 5 (bottleneck_chain.hip, C3): conv2 of the Bottlenecks without a projection shortcut computed in front of conv3 inside
    the chain kernel -- against the same arithmetic as separate launches (HRN_DISABLE_CHAIN3=1: conv_direct_kernel + chain kernel):
    the stored tensors behind it (conv3 of every block, conv1 of the next) and the heat-maps, bit for bit; ragged micro-batches;
    the last block of the layer (no conv1 behind it); pad positions stay zero."""
    crops = torch.from_numpy(pkg.synth_crops(n, h, w, seed=91)).cuda()
    boxes = pkg.synth_boxes(n, seed=92)
    outs = []
    for on in (True, False):
        monkeypatch.delenv("HRN_DISABLE_CHAIN3", raising=False)
        if not on:
            monkeypatch.setenv("HRN_DISABLE_CHAIN3", "1")
        net = pkg.NativeHRNet(c, 17, (h, w), "bf16", max_batch=mb, device=0, model_name=model).load_state_dict(
            pkg.synth_state_dict(c, 17, 3, model=model))
        tapped = {t.name.decode() for t in net.tap_infos()}
        l1 = sorted(t for t in tapped if t.startswith("layer1."))
        nb = 4 if model == "HRNet" else 3
        assert (("layer1.1.conv2" in tapped), ("layer1.%d.conv2" % (nb - 1) in tapped), ("layer1.0.conv2" in tapped)) == ((not on), (not on), True)
        hm, pts = net.predict_crops(crops, boxes, return_heatmaps=True)
        x1 = crops[:min(n, mb)].contiguous()
        taps = {t: net.forward_tap(x1, t).cpu().numpy() for t in l1 if t.endswith("conv3") or t.endswith("conv1")}
        assert net.pad_violations() == 0
        outs.append((hm.cpu().numpy(), pts.cpu().numpy(), taps))
        net.close()
    assert np.isfinite(outs[0][0]).all() and np.abs(outs[0][0]).max() > 0
    for t, v in outs[0][2].items():
        np.testing.assert_array_equal(v, outs[1][2][t], err_msg=t)
        assert np.abs(v).max() > 0, t
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("c,h,w,n", [(48, 128, 96, 96), (32, 128, 96, 128), (48, 256, 192, 40)])
def test_generic_kernel_weights_through_lds_is_bit_identical(pkg, monkeypatch, c, h, w, n):
    """kernels.hip WL: the generic conv kernel with its weights staged through LDS (full-size tiles only, hence the
    large batches) against one copy per wave straight from L2: same bits."""
    crops = torch.from_numpy(pkg.synth_crops(n, h, w, seed=51)).cuda()
    boxes = pkg.synth_boxes(n, seed=52)
    outs = []
    for on in ("1", "0"):
        monkeypatch.setenv("HRN_DIRECT_WLDS", on)
        net = _engine(pkg, c, h, w, "bf16", max_batch=n, seed=4)
        hm, pts = net.predict_crops(crops, boxes, return_heatmaps=True)
        outs.append((hm.cpu().numpy(), pts.cpu().numpy()))
        net.close()
    assert np.isfinite(outs[0][0]).all()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


def test_bf16_head_mfma_vs_fp32_weight_head(pkg, monkeypatch):
    """The MFMA head rounds final_layer's weights to bf16; the VALU head keeps them fp32.  Same activations in both:
    the heat-maps may differ by the weight rounding only (2^-9 relative per term)."""
    c, h, w, n = 32, 128, 96, 3
    crops = torch.from_numpy(pkg.synth_crops(n, h, w, seed=31)).cuda()
    boxes = pkg.synth_boxes(n, seed=32)
    monkeypatch.delenv("HRN_DISABLE_HEAD_MFMA", raising=False)
    a = _engine(pkg, c, h, w, "bf16", max_batch=4, seed=5)
    hm_a, pts_a = a.predict_crops(crops, boxes, return_heatmaps=True)
    only_pts = a.predict_crops(crops, boxes)                      # heat-map write-out off: same arg-max path
    assert torch.equal(only_pts, pts_a)
    monkeypatch.setenv("HRN_DISABLE_HEAD_MFMA", "1")
    b = _engine(pkg, c, h, w, "bf16", max_batch=4, seed=5)
    hm_b, pts_b = b.predict_crops(crops, boxes, return_heatmaps=True)
    hm_a, hm_b = hm_a.cpu().numpy(), hm_b.cpu().numpy()
    scale = np.abs(hm_b).max()
    assert np.abs(hm_a - hm_b).max() <= 0.01 * scale + 1e-3
    # arg-max of the MFMA head is the first maximum of ITS OWN heat-maps (decode on the GPU == numpy on the host)
    flat = hm_a.reshape(n, 17, -1)
    idx = flat.argmax(-1)
    T = _oracle()
    ref = T.decode_heatmaps(hm_a, boxes)
    np.testing.assert_array_equal(pts_a.cpu().numpy()[..., :2], ref[..., :2])
    assert idx.shape == (n, 17)
    a.close(), b.close()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_other_joint_count_and_two_engines_on_side_streams(pkg, dtype):
    """MPII-style 16 joints (the head pads joints to two MFMA fragments), two engines alive at once, each driven
    from its own non-default stream: results must equal the oracle (fp32) / the fp32 engine within the bf16 bound."""
    T = _oracle()
    c, h, w, n, joints = 32, 128, 96, 3, 16
    sd_np = pkg.synth_state_dict(c, joints, 9)
    crops = torch.from_numpy(pkg.synth_crops(n, h, w, seed=41)).cuda()
    boxes = pkg.synth_boxes(n, seed=42)
    ref_hm, ref_pts = T.predict_crops(pkg.synth.to_torch_state_dict(sd_np), crops.cpu(), boxes)
    a = pkg.NativeHRNet(c, joints, (h, w), dtype, max_batch=2, device=0).load_state_dict(sd_np)
    b = pkg.NativeHRNet(c, joints, (h, w), dtype, max_batch=4, device=0).load_state_dict(sd_np)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for _ in range(3):  # interleave the two engines on two streams
        with torch.cuda.stream(sa):
            ha, pa = a.predict_crops(crops, boxes, return_heatmaps=True)
        with torch.cuda.stream(sb):
            hb, pb = b.predict_crops(crops, boxes, return_heatmaps=True)
        outs.append((ha, pa, hb, pb))
    torch.cuda.synchronize()
    for ha, pa, hb, pb in outs:
        assert tuple(ha.shape) == (n, joints, h // 4, w // 4) and tuple(pa.shape) == (n, joints, 3)
        assert torch.equal(ha, hb) and torch.equal(pa, pb)          # micro-batch 2 vs 4, stream a vs b
        assert torch.equal(ha, outs[0][0]) and torch.equal(pa, outs[0][1])
    hm, pts = outs[0][0].cpu().numpy(), outs[0][1].cpu().numpy()
    if dtype == "fp32":
        np.testing.assert_allclose(hm, ref_hm, rtol=0, atol=HM_ATOL_F32)
        np.testing.assert_array_equal(pts[..., :2], ref_pts[..., :2])
    else:
        err = np.abs(hm - ref_hm).max()
        assert err < 0.05 * ref_hm.std() + 0.05, err
        np.testing.assert_array_equal(pts[..., :2], T.decode_heatmaps(hm, boxes)[..., :2])
    a.close(), b.close()


def test_predict_stream_hides_uploads_and_matches_predict_crops(pkg):
    """Host-resident batches through the double-buffered upload pipeline: same results as predict_crops, ragged sizes."""
    c, h, w = 32, 128, 96
    net = _engine(pkg, c, h, w, "bf16", max_batch=8, seed=2)
    sizes = [8, 3, 8, 1, 5]
    items = []
    for k, n in enumerate(sizes):
        x = torch.from_numpy(pkg.synth_crops(n, h, w, seed=60 + k))
        items.append((x.pin_memory(), pkg.synth_boxes(n, seed=70 + k)))
    outs = [(hm.cpu().numpy(), pts.cpu().numpy()) for hm, pts in net.predict_stream(iter(items), return_heatmaps=True)]
    assert len(outs) == len(sizes)
    for (x, b), (hm, pts) in zip(items, outs):
        ref_hm, ref_pts = net.predict_crops(x.cuda(), b, return_heatmaps=True)
        np.testing.assert_array_equal(hm, ref_hm.cpu().numpy())
        np.testing.assert_array_equal(pts, ref_pts.cpu().numpy())
    assert list(net.predict_stream(iter([]))) == []
    with pytest.raises(ValueError):
        list(net.predict_stream([(torch.zeros((9, 3, h, w)), pkg.synth_boxes(9))]))
    # round 5: uint8 BGR crops at the network's resolution (a quarter of the PCIe bytes; colour flip + ToTensor + Normalize on the GPU)
    # -- mixed with fp32 batches in one stream, against the reference transform's float32 arithmetic done on the host
    rng = np.random.default_rng(11)
    items8, refs = [], []
    for k, n in enumerate([8, 2, 5]):
        u8 = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)                       # BGR, HWC (what cv2.resize leaves)
        rgb = u8[..., ::-1].astype(np.float32) / np.float32(255.0)                       # cvtColor + ToTensor (SimpleHRNet.py:218-221, 167-172)
        x = ((rgb - np.array([0.485, 0.456, 0.406], np.float32)) / np.array([0.229, 0.224, 0.225], np.float32)).transpose(0, 3, 1, 2)
        b = pkg.synth_boxes(n, seed=90 + k)
        items8.append((torch.from_numpy(u8).pin_memory(), b))
        refs.append(net.predict_crops(torch.from_numpy(np.ascontiguousarray(x)).cuda(), b).cpu().numpy())
        if k == 1:   # an fp32 batch in between: the staging buffers follow the dtype of what arrives
            items8.append(items[0]), refs.append(outs[0][1])
    got = [pts.cpu().numpy() for pts in net.predict_stream(iter(items8))]
    for g, r in zip(got, refs):
        np.testing.assert_array_equal(g, r)
    with pytest.raises(ValueError):
        list(net.predict_stream([(torch.zeros((2, h, w + 1, 3), dtype=torch.uint8), pkg.synth_boxes(2))]))
    with pytest.raises(ValueError):   #  a wrong-shaped fp32 batch is refused by name, not by a broadcast error inside copy_
        list(net.predict_stream([(torch.zeros((2, 3, h, w + 1)), pkg.synth_boxes(2))]))
    # the uint8 form normalises into ONE fp32 buffer per staging slot (resize_frames(out=...)): a caller-supplied buffer is validated
    buf = torch.empty((4, 3, h, w), dtype=torch.float32, device="cuda")
    u8 = torch.randint(0, 256, (3, h, w, 3), dtype=torch.uint8)
    a = net.resize_frames(u8, 0, out=buf)
    assert a.data_ptr() == buf.data_ptr() and torch.equal(a, net.resize_frames(u8, 0))
    with pytest.raises(ValueError):
        net.resize_frames(u8, 0, out=torch.empty((2, 3, h, w), dtype=torch.float32, device="cuda"))
    net.close()
