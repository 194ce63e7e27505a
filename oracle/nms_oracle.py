"""TEST INFRASTRUCTURE ONLY -- restatement of the reference's pure-python greedy NMS (misc/nms/nms.py:35-72), the CPU
twin of its CUDA kernel (misc/nms/nms_kernel.cu:20-31 `devIoU`, same +1 pixel convention, same `> thresh` rule).
Pinned by tests/golden/nms_cases.npz, which was produced by the reference function itself."""
import numpy as np


def nms(dets: np.ndarray, thresh: float):
    """dets (n, >=5) [x1, y1, x2, y2, score] -> indices kept, highest score first; a box survives while its overlap
    with every kept box is <= thresh."""
    if dets.shape[0] == 0:
        return []
    x1, y1, x2, y2, scores = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3], dets[:, 4]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = scores.argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        rest = order[1:]
        w = np.maximum(0.0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + 1)
        h = np.maximum(0.0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + 1)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        order = rest[np.where(ovr <= thresh)[0]]
    return keep
