"""Per-block phase timing of the grouped conv3x3 kernel (needs a build with -DHRN_C3_TIMING)."""
import ctypes, importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib_mod = importlib.import_module("simple-hrnet_amd._lib")
lib_mod.HIPCC_FLAGS.append("-DHRN_C3_TIMING")
for extra in os.environ.get("C3_DEFS", "").split():
    lib_mod.HIPCC_FLAGS.append("-D" + extra)
lib_mod.LIB_PATH = lib_mod.LIB_PATH.replace(".so", "_timing.so")
if not os.environ.get("NO_BUILD"):   # NO_BUILD=1: the _timing.so was built beforehand (no GPU needed for that)
    lib_mod.build(force=True)
import torch
pkg = importlib.import_module("simple-hrnet_amd")
mb = int(os.environ.get("MB", "64"))
net = pkg.NativeHRNet(48, 17, (384, 288), "bf16", max_batch=mb, device=0).load_state_dict(pkg.synth_state_dict(48, 17, 0))
lib = lib_mod.load()
lib.hrn_debug_c3_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
NB = 32768
buf = np.zeros((NB, 8, 8), np.int64)
lib.hrn_debug_c3_timing(buf.ctypes.data, NB)          # allocate + arm
x = torch.randn((mb, 3, 384, 288), device="cuda")
for _ in range(3):
    net(x)
torch.cuda.synchronize()
lib.hrn_debug_c3_timing(buf.ctypes.data, NB)
n96 = buf[(buf[:, 0, 6] >= 90) & (buf[:, 0, 6] < 100)]   # 96-cout form (conv3x3_n96.inc): per-stage / per-tile phases
if len(n96):
    for S in sorted(set(n96[:, 0, 5] // np.maximum(1, n96[:, 0, 7] >> 48) // 3)):
        sel = n96[(n96[:, 0, 5] // np.maximum(1, n96[:, 0, 7] >> 48) // 3) == S]
        st, nt = sel[:, :, 5].astype(float), (sel[:, :, 7] >> 48).astype(float)
        erest = (sel[:, :, 7] & ((1 << 48) - 1)).astype(float)
        print("96-cout form, S=%d (cin %d): %d blocks, %.1f stages, %.1f tiles per block, total/block %.0f ticks (ideal MFMA per stage and SIMD: %d)" % (
            S, 32 * S, len(sel), st[:, 0].mean(), nt[:, 0].mean(), sel[:, 0, 4].mean(), 2 * 18 * (int(sel[0, 0, 6]) - 90) * 16))
        plan = (sel[:, :, 1] & 0xffffffff).astype(float)
        pro, pro_issue = (sel[:, :, 1] >> 32).astype(float), (sel[:, :, 3] >> 32).astype(float)
        ewait = (sel[:, :, 3] & 0xffffffff).astype(float)
        print("    block prologue (entry -> first stage): %.0f ticks, of which setup + LDS-DMA issue %.0f" % (pro[:, 0].mean(), pro_issue[:, 0].mean()))
        rt = (sel[:, 0, 0] >> 32).astype(float)   # the block's life in 100-MHz ticks (s_memrealtime)
        print("    shader clock while these blocks ran: %.0f MHz (s_memtime / s_memrealtime)" % (100.0 * sel[:, 0, 4].sum() / max(1.0, rt.sum())))
        for w in range(8):
            print("    wave %d: per stage: entry wait %.0f  plan/residual issue %.0f  chunks %.0f | per tile: residual wait %.0f  epilogue rest %.0f | prologue %.0f (issue %.0f)" % (
                w, ((sel[:, w, 0] & 0xffffffff) / st[:, w]).mean(), (plan[:, w] / st[:, w]).mean(), (sel[:, w, 2] / st[:, w]).mean(),
                (ewait[:, w] / nt[:, w]).mean(), (erest[:, w] / nt[:, w]).mean(), pro[:, w].mean(), pro_issue[:, w].mean()))
    buf[(buf[:, 0, 6] >= 90) & (buf[:, 0, 6] < 100)] = 0
fz = buf[buf[:, 0, 6] >= 100]   # fused BasicBlock blocks (bbf_run) record their own phases
if len(fz):
    nt = fz[:, 0, 5].astype(float)
    print("fused BasicBlock blocks: %d, %.1f tiles/block, total/tile %.0f ticks" % (len(fz), nt.mean(), (fz[:, 0, 4] / nt).mean()))
    for w in range(8):
        per = lambda col: (fz[:, w, col] / nt).mean()
        post = ((fz[:, w, 7] & 0xffffffff) / nt).mean()
        epi = ((fz[:, w, 7] >> 32) / nt).mean()
        print("    wave %d (%d frags): top wait %.0f  conv1 loop %.0f  conv1 tail %.0f  conv2 loop %.0f  refill %.0f  epilogue %.0f" % (
            w, fz[0, w, 6] - 100, per(0), per(1), per(2), per(3), post, epi))
    buf[buf[:, 0, 6] >= 100] = 0
b = buf[buf[:, 0, 5] > 0]
print("blocks of the last grouped launch:", len(b))
for mr in (2, 4, 8):
    for S in sorted(set(b[:, 0, 7])):
        sel = b[(b[:, 0, 6] == mr) & (b[:, 0, 7] == S)]
        if not len(sel): continue
        nh = sel[:, 0, 5].mean()
        print("MR=%d S=%d: %4d blocks, %.1f half-stages/block, total/block %.0f, epilogue/tile %.0f (ideal mfma %d/half/wave)" % (
            mr, S, len(sel), nh, sel[:, 0, 4].mean(), (sel[:, 0, 3] / np.maximum(1, sel[:, 0, 5] / (2 * S))).mean(), 7 * 3 * mr * 16))
        for w in range(8):
            print("    wave %d: wait %.0f issue %.0f compute %.0f" % (w, (sel[:, w, 0] / sel[:, w, 5]).mean(), (sel[:, w, 1] / sel[:, w, 5]).mean(), (sel[:, w, 2] / sel[:, w, 5]).mean()))
