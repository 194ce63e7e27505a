"""Host-side logic of the native library on a CPU-only box: the C ABI loads and exports every symbol
the header declares, the graph compiler reproduces the reference's op census and FLOP count, BN folding
and MFMA-fragment packing are correct.  No compute entry point is called (there is no CPU compute path)."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_pkg, state_dict_np

pkg = load_pkg()
_lib = load_pkg("_lib")


def test_library_exports_every_header_symbol():
    lib = _lib.load()
    declared = _lib.header_symbols()
    assert set(declared) == set(_lib.SYMBOLS), (declared, sorted(_lib.SYMBOLS))
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.hrn_version()


@pytest.mark.parametrize("c,res,gflop", [(32, (256, 192), 15.290), (48, (384, 288), 70.613)])
def test_plan_census_matches_survey(c, res, gflop):
    net = pkg.NativeHRNet(c, 17, res, "bf16", max_batch=4, device=-1)
    infos = net.conv_infos()
    # 293 Conv2d in the reference; conv1 (stem kernel) and final_layer (head kernel) are not
    # generic-conv ops here
    assert len(infos) == 291
    assert abs(net.flops_per_crop() / 1e9 - gflop) < 0.002 * gflop
    names = [i.name.decode() for i in infos]
    spec = {k[:-len(".weight")] for k, s, kind in pkg.synth.hrnet_state_spec(c, 17) if kind == "conv"}
    assert set(names) == spec - {"conv1", "final_layer"}
    # hot convs: 208 BasicBlock 3x3 s1 convs, each 0.28665 GFLOP at W48 384x288
    hot = [i for i in infos if ".branches." in i.name.decode()]
    assert len(hot) == 208
    if c == 48:
        assert all(abs(i.flops / 1e9 - 0.28665) < 1e-4 for i in hot)
    net.close()


def test_bad_arguments_raise():
    with pytest.raises(ValueError):
        pkg.NativeHRNet(48, 17, (384, 280), "bf16", device=-1)      # not a multiple of 32
    with pytest.raises(ValueError):
        pkg.NativeHRNet(40, 17, (384, 288), "bf16", device=-1)      # c not a multiple of 16
    with pytest.raises(ValueError):
        pkg.NativeHRNet(48, 17, (384, 288), "int8", device=-1)
    with pytest.raises(ValueError):
        pkg.NativeHRNet(48, 17, (384, 288), "bf16", device="cpu")   # 'Wrong device name.' like SimpleHRNet.py:139
    net = pkg.NativeHRNet(32, 17, (64, 64), "fp32", max_batch=2, device=-1)
    sd = dict(state_dict_np(32))
    del sd["stage3.2.branches.1.0.bn1.running_var"]
    with pytest.raises(KeyError, match="stage3.2.branches.1.0.bn1.running_var"):
        net.load_state_dict(sd)
    # a plan-only handle must refuse to compute: there is NO CPU fallback
    net.load_state_dict(state_dict_np(32))
    rc = net._lib.hrn_forward(net._h, ctypes.c_void_p(16), 1, None, 0, None, ctypes.c_void_p(16), None)
    assert rc != 0 and b"no CPU compute path" in net._lib.hrn_last_error(net._h)
    net.close()


def _unpack_lds(net, info):
    """inverse of the slice-major image of the LDS-staged 3x3 kernel:
    [cout tile][slice][chunk][frag][lane][8 bf16], k_local = tap*ks + ci_local per slice"""
    raw = net.read_blob(info.w_offset, info.w_bytes)
    ks, nrb = info.ks, info.nr
    if ks == 16:   # the fp32 form (conv3x3_f32.hip): [cout tile][slice][9 taps][frag][lane][4 fp32], k = 4 g + e
        slices, ntiles = info.cin // 16, info.cout // (16 * nrb)
        vals = raw.view(np.float32).reshape(ntiles, slices, 9, nrb, 64, 4)
        out = np.zeros((info.cout, 9 * info.cin), np.float32)
        for lane in range(64):
            li, g = lane & 15, lane >> 4
            for t in range(ntiles):
                for j in range(nrb):
                    co = t * 16 * nrb + (li >> 2) * 4 * nrb + j * 4 + (li & 3)
                    for s in range(slices):
                        for c in range(9):
                            out[co, c * info.cin + s * 16 + 4 * g:c * info.cin + s * 16 + 4 * g + 4] = vals[t, s, c, j, lane]
        return out
    vals = (raw.view(np.uint16).astype(np.uint32) << 16).view(np.float32)
    slices, ntiles, nch = info.cin // ks, info.cout // (16 * nrb), (9 * ks + 31) // 32
    vals = vals.reshape(ntiles, slices, nch, nrb, 64, 8)
    out = np.zeros((info.cout, 9 * info.cin), np.float32)
    pad = []
    for t in range(ntiles):
        for s in range(slices):
            for c in range(nch):
                for j in range(nrb):
                    for lane in range(64):
                        li, g = lane & 15, lane >> 4
                        co = t * 16 * nrb + (li >> 2) * 4 * nrb + j * 4 + (li & 3)
                        if info.algo == 3:   # the 96-cout form: 8 contiguous channels per lane in each 32-channel group
                            co = t * 96 + (j >> 1) * 32 + (li >> 2) * 8 + (j & 1) * 4 + (li & 3)
                        for e in range(8):
                            kl = 32 * c + 8 * g + e
                            v = vals[t, s, c, j, lane, e]
                            if kl < 9 * ks:
                                out[co, (kl // ks) * info.cin + s * ks + kl % ks] = v
                            else:
                                pad.append(v)
    assert not np.any(pad)
    return out


def _unpack(net, info, dtype):
    """inverse of the documented fragment-major layout"""
    if info.algo in (1, 3):   # 3 = the 96-cout form: same image with ks = 32, nr = 6
        return _unpack_lds(net, info)
    kc, vec = (32, 8) if dtype == "bf16" else (16, 4)
    raw = net.read_blob(info.w_offset, info.w_bytes)
    if dtype == "bf16":
        u = raw.view(np.uint16).astype(np.uint32) << 16
        vals = u.view(np.float32)
    else:
        vals = raw.view(np.float32)
    kchunks = info.kpad // kc
    vals = vals.reshape(info.cout // 16, kchunks, 64, vec)
    out = np.zeros((info.cout, info.kpad), np.float32)
    for f in range(info.cout // 16):
        ng, j = divmod(f, info.nr)
        for lane in range(64):
            li, g = lane & 15, lane >> 4
            co = ng * 16 * info.nr + (li >> 2) * 4 * info.nr + j * 4 + (li & 3)
            for k in range(kchunks):
                out[co, k * kc + g * vec:k * kc + (g + 1) * vec] = vals[f, k, lane]
    return out


def _bf16_round(a):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return t.to(torch.bfloat16).to(torch.float32).numpy()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_fold_and_pack(dtype):
    c = 48
    sd = state_dict_np(c)
    net = pkg.NativeHRNet(c, 17, (64, 64), dtype, max_batch=1, device=-1).load_state_dict(sd)
    infos = {i.name.decode(): i for i in net.conv_infos()}
    cases = [("conv2", "bn2"), ("layer1.0.conv1", "layer1.0.bn1"), ("transition1.0.0", "transition1.0.1"),
             ("stage3.1.branches.1.3.conv2", "stage3.1.branches.1.3.bn2"),
             ("stage4.0.fuse_layers.3.0.2.0", "stage4.0.fuse_layers.3.0.2.1"),
             ("stage4.2.fuse_layers.0.3.0", "stage4.2.fuse_layers.0.3.1")]
    for conv, bn in cases:
        info = infos[conv]
        w = sd[conv + ".weight"].astype(np.float64)
        scale = sd[bn + ".weight"].astype(np.float64) / np.sqrt(sd[bn + ".running_var"].astype(np.float64) + 1e-5)
        shift = sd[bn + ".bias"].astype(np.float64) - sd[bn + ".running_mean"].astype(np.float64) * scale
        cout, cin, kh, kw = w.shape
        assert (info.cout, info.cin, info.ksize) == (cout, cin, kh)
        want = (w * scale[:, None, None, None]).transpose(0, 2, 3, 1).reshape(cout, kh * kw * cin)  # k = tap*cin+ci
        want = want.astype(np.float32)
        if dtype == "bf16":
            want = _bf16_round(want)
        got = _unpack(net, info, dtype)
        np.testing.assert_array_equal(got[:, :want.shape[1]], want)
        assert not got[:, want.shape[1]:].any()          # K padding is zero
        bias = net.read_blob(info.b_offset, 4 * cout).view(np.float32)
        np.testing.assert_allclose(bias, shift.astype(np.float32), rtol=0, atol=0)
    net.close()


@pytest.mark.parametrize("res,n", [((256, 192), 256), ((256, 192), 150), ((128, 96), 256), ((64, 64), 7)])
def test_block_maps_cover_every_tile_at_other_resolutions(res, n, monkeypatch):
    """same invariant at other row pitches, with the fused BasicBlock pass forced on for every call size"""
    monkeypatch.setenv("HRN_BBF_MIN_TILES", "1")
    net = pkg.NativeHRNet(48, 17, res, "bf16", max_batch=256, device=-1)
    infos = net.conv_infos()
    blocks, members = np.zeros((40000, 6), np.int32), np.zeros(64, np.int32)
    group, fused_blocks = 0, 0
    while True:
        nb = net._lib.hrn_plan_block_map(net._h, group, n, group & 1, blocks.ctypes.data, len(blocks), members.ctypes.data, len(members))
        if nb < 0:
            break
        covered = {}
        for d, nt, tiles, mt0, px, flags in blocks[:nb]:
            conv = int(members[d]) & ~(1 << 30)
            i = infos[conv]
            mtiles = -(-n * (i.out_h * i.out_w if net.conv_compact(conv) else (i.out_h + 1) * (i.out_w + 1)) // px)
            cov = covered.setdefault((conv, int(px)), np.zeros((i.cout // (16 * i.nr), mtiles), np.int32))
            cov[nt, mt0:min(mtiles, mt0 + tiles)] += 1
            fused_blocks += int(flags & 1)
        assert all((cov == 1).all() for cov in covered.values())
        group += 1
    assert group == 66 and fused_blocks > 0
    net.close()


@pytest.mark.parametrize("n", [1, 3, 8, 12, 20, 64, 96, 200, 256])   # (8, 12: small launches in which the short convolutions keep full tiles, round 6)
@pytest.mark.parametrize("reverse", [0, 1])
def test_block_maps_cover_every_tile_exactly_once(n, reverse):
    """The block maps of the grouped BasicBlock launches (hrnet_mi355.cpp: group_blocks) for W48 384x288: whatever the
    block lengths, tile sizes (512 / 384 / 128 pixels) and order chosen for a call of n crops, every (conv, cout tile, M
    tile) is produced exactly once -- by its own launch, or for the 48-channel BasicBlocks by the fused pass of conv1's
    launch, in which case conv2's launch must not touch it."""
    net = pkg.NativeHRNet(48, 17, (384, 288), "bf16", max_batch=256, device=-1)
    infos = net.conv_infos()
    names = [i.name.decode() for i in infos]
    lib = net._lib
    blocks = np.zeros((40000, 6), np.int32)
    members = np.zeros(64, np.int32)
    fused_conv1, plain_done, group = set(), {}, 0
    while True:
        nb = lib.hrn_plan_block_map(net._h, group, n, reverse, blocks.ctypes.data, len(blocks), members.ctypes.data, len(members))
        if nb < 0:
            break
        assert 0 < nb <= len(blocks)
        covered = {}
        for d, nt, tiles, mt0, px, flags in blocks[:nb]:
            conv, fused = int(members[d]) & ~(1 << 30), bool(int(members[d]) >> 30)
            i = infos[conv]
            assert fused == bool(flags & 1) and (px == 512 if fused else px in (512, 384, 128)) and ((flags & 2) != 0) == (px == 128)
            mtiles = -(-n * (i.out_h * i.out_w if net.conv_compact(conv) else (i.out_h + 1) * (i.out_w + 1)) // px)
            assert 0 <= nt < i.cout // (16 * i.nr) and tiles >= 1 and 0 <= mt0 < mtiles
            cov = covered.setdefault((conv, fused, int(px)), np.zeros((i.cout // (16 * i.nr), mtiles), np.int32))
            cov[nt, mt0:min(mtiles, mt0 + tiles)] += 1
        for (conv, fused, px), cov in covered.items():
            assert (cov == 1).all(), (names[conv], px, np.unique(cov))
            assert conv not in plain_done, names[conv]            # one tile size, one form per conv and call
            plain_done[conv] = fused
            if fused:
                fused_conv1.add(names[conv])
        group += 1
    assert group == 66                                            # 64 BasicBlock launches + layer1.0's 3x3 + transition1 (round 5: the other 3x3s of layer1 run inside the chain kernel)
    hot = [k for k, nm in enumerate(names) if ".branches." in nm]
    big = -(-n * 97 * 73 // 512) >= 1100                          # hrn_ctx::bbf_min_tiles
    for k in hot:
        nm = names[k]
        if nm.endswith("conv2") and nm.replace("conv2", "conv1") in fused_conv1:
            assert k not in plain_done, nm                        # computed by conv1's fused pass: no blocks of its own
        else:
            assert k in plain_done, nm
    assert len(fused_conv1) == (32 if big else 0)
    net.close()


@pytest.mark.parametrize("c,res,dtype", [(48, (384, 288), "bf16"), (32, (256, 192), "fp32")])
@pytest.mark.parametrize("n", [1, 5, 64, 256])
def test_direct_group_maps_cover_every_tile_exactly_once(c, res, dtype, n):
    """Grouped launches of the generic kernel (hrnet_mi355.cpp: direct_group_blocks): every (conv, cout tile, M tile) once;
    entries beyond the last M tile are the XCD-alignment padding."""
    net = pkg.NativeHRNet(c, 17, res, dtype, max_batch=256, device=-1)
    infos = net.conv_infos()
    blocks = np.zeros((60000, 3), np.int32)
    members = np.zeros(256, np.int32)
    px = ctypes.c_int32(0)
    group, seen = 0, set()
    while True:
        nb = net._lib.hrn_plan_direct_map(net._h, group, n, blocks.ctypes.data, len(blocks), members.ctypes.data, len(members),
                                          ctypes.byref(px))
        if nb < 0:
            break
        assert 0 < nb <= len(blocks) and px.value in (64, 128, 256)
        cov = {}
        for d, ng, mt in blocks[:nb]:
            i = infos[members[d]]
            grid_h, grid_w = (i.out_h, i.out_w)
            mtiles = -(-n * (grid_h + 1) * (grid_w + 1) // px.value)
            assert 0 <= ng < i.cout // (16 * i.nr) and mt >= 0
            if mt < mtiles:
                cov.setdefault(int(members[d]), np.zeros((i.cout // (16 * i.nr), mtiles), np.int32))[ng, mt] += 1
        for conv, m in cov.items():
            assert (m == 1).all(), (infos[conv].name, np.unique(m))
            assert conv not in seen
            seen.add(conv)
        group += 1
    assert group >= 8 and len(seen) >= 2 * group
    net.close()
