"""Per-CU timeline of the last grouped BasicBlock launches of a pass (library built with -DHRN_Q_TIMING: every block records the CU it
ran on and its start / end on the 100-MHz s_memrealtime clock).  For each of the last launches: span, mean CU busy share, when the
first CU runs out of work, block durations by length class.   NO_BUILD=1: the _qtiming.so was built beforehand."""
import ctypes, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib_mod = importlib.import_module("simple-hrnet_amd._lib")
lib_mod.HIPCC_FLAGS.append("-DHRN_Q_TIMING")
lib_mod.LIB_PATH = lib_mod.LIB_PATH.replace(".so", "_qtiming.so")
if not os.environ.get("NO_BUILD"):
    lib_mod.build(force=True)
import torch
pkg = importlib.import_module("simple-hrnet_amd")
mb = int(os.environ.get("MB", "256"))
net = pkg.NativeHRNet(48, 17, (384, 288), "bf16", max_batch=mb, device=0).load_state_dict(pkg.synth_state_dict(48, 17, 0))
lib = lib_mod.load()
lib.hrn_debug_q_timing.argtypes = [ctypes.c_void_p]
SLOTS, NB = 4, 8192
buf = np.zeros((SLOTS, NB, 4), np.int64)
lib.hrn_debug_q_timing(buf.ctypes.data)
x = torch.randn((mb, 3, 384, 288), device="cuda")
for _ in range(3):
    net(x)
torch.cuda.synchronize()
lib.hrn_debug_q_timing(buf.ctypes.data)
order = sorted(range(SLOTS), key=lambda s: int(buf[s, 0, 1] >> 32))
prev_end = None
for s in order:
    grid, seq = int(buf[s, 0, 1] & 0xffffffff), int(buf[s, 0, 1] >> 32)
    if grid <= 0:
        continue
    r = buf[s, :min(grid, NB)]
    r = r[(r[:, 1] >> 32) == seq]
    t0, t1, cu = r[:, 0], r[:, 2], r[:, 3]
    base = t0.min()
    if prev_end is not None:
        print("  -- %.2f us between the last block end of the previous launch and the first block start of this one" % ((base - prev_end) / 100.0))
    prev_end = t1.max()
    span = (t1.max() - base) / 100.0
    dur = (t1 - t0) / 100.0
    cus = np.unique(cu)
    busy = np.array([dur[cu == c].sum() for c in cus])
    last_end = np.array([(t1[cu == c].max() - base) / 100.0 for c in cus])
    first_start = np.array([(t0[cu == c].min() - base) / 100.0 for c in cus])
    # idle inside a CU's life: gaps between consecutive blocks
    gaps = []
    for c in cus:
        o = np.argsort(t0[cu == c])
        a, b = t0[cu == c][o], t1[cu == c][o]
        gaps.append(((a[1:] - b[:-1]).clip(min=0).sum()) / 100.0)
    gaps = np.array(gaps)
    print("launch seq %d: %d blocks on %d CUs, span %.1f us | CU busy %.1f us mean (%.1f %% of span; min %.1f max %.1f) | gaps between blocks %.2f us per CU | "
          "first block starts %.1f..%.1f us | a CU's last block ends: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us" % (
              seq, len(r), len(cus), span, busy.mean(), 100 * busy.mean() / span, busy.min(), busy.max(), gaps.mean(), first_start.min(), first_start.max(),
              last_end.min(), np.percentile(last_end, 10), np.median(last_end), np.percentile(last_end, 90), last_end.max()))
    # durations by class (rounded to 5 us) with start-time ranges
    step = float(os.environ.get('STEP', '5'))
    cls = np.round(dur / step) * step
    for d in sorted(set(cls))[::-1][:int(os.environ.get('CLASSES', '12'))]:
        m = cls == d
        print("    blocks of ~%5.1f us: %4d, started %.1f..%.1f us, ended %.1f..%.1f us" % (d, m.sum(), ((t0[m] - base) / 100.0).min(), ((t0[m] - base) / 100.0).max(),
                                                                                      ((t1[m] - base) / 100.0).min(), ((t1[m] - base) / 100.0).max()))
