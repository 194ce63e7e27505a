"""``gpu_nms`` of the reference (misc/nms/gpu_nms.pyx:19-34) on the MI355X library: same signature, same result.

The Cython original sorts by score on the host, hands the sorted boxes to the native ``_nms`` and maps the kept rows
back; so does this, with ``hrn_nms`` (csrc/nms.hip) as the native part.  No CPU fallback: without a GPU it raises."""
from __future__ import annotations

import ctypes
from typing import List

import numpy as np

from . import _lib


def gpu_nms(dets: np.ndarray, thresh: float, device_id: int = 0) -> List[int]:
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    if dets.ndim != 2 or dets.shape[1] < 5:
        raise ValueError("dets must be (n, >=5): x1, y1, x2, y2, score")
    n, dim = dets.shape
    by_score = dets[:, 4].argsort()[::-1].astype(np.int32)          # the native part wants score-sorted rows
    rows = np.ascontiguousarray(dets[by_score])
    kept_rows = np.empty(n, dtype=np.int32)
    count = ctypes.c_int32(0)
    lib = _lib.load()
    rc = lib.hrn_nms(kept_rows.ctypes.data, ctypes.byref(count), rows.ctypes.data, n, dim, ctypes.c_float(thresh), int(device_id))
    if rc != 0:
        raise (ValueError if rc in (1, 2) else RuntimeError)("hrn_nms failed: " + lib.hrn_nms_last_error().decode())
    return list(by_score[kept_rows[:count.value]])                   # back to the caller's row numbers
