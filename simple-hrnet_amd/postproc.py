"""The reference's host-side pose post-processing on the native library: OKS NMS (``misc/nms/nms.py:97-180``) and the
tracker's association step (``misc/utils.py:372-429``) -- same names, arguments and results.

The arithmetic (OKS, box IoU, the similarity matrices, the optimal assignment) is in ``csrc/postproc.cpp`` behind the C
ABI; what the reference does with numpy around it -- the score sort, the float32 blend of the two similarity matrices,
thresholding, temporal smoothing (with numpy's casting into the caller's arrays) and the numbering of new people -- is
done with the same numpy expressions here, so dtypes and tie-breaks are numpy's.  No pure-Python twin: without the
library these functions raise like the rest of the package."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import numpy as np

from . import _lib


def _nan_if_none(v) -> float:
    return float("nan") if v is None else float(v)


def _db_arrays(kpts_db: Sequence[dict]):
    entries = list(kpts_db)
    scores = np.array([e["score"] for e in entries])
    kpts = np.ascontiguousarray(np.stack([np.asarray(e["keypoints"]).reshape(-1) for e in entries]), np.float64)
    areas = np.ascontiguousarray(np.array([e["area"] for e in entries]), np.float64)
    if kpts.ndim != 2 or kpts.shape[1] % 3:
        raise ValueError("keypoints must be (J, 3) per entry")
    return scores, kpts, areas


def _sigmas(sigmas, joints):
    if isinstance(sigmas, np.ndarray):
        s = np.ascontiguousarray(sigmas, np.float64)
        if s.shape != (joints,):
            raise ValueError("sigmas must have one value per joint")
        return s
    if joints != 17:
        raise ValueError("the default sigmas are COCO's 17; pass sigmas for %d joints" % joints)
    return None


def oks_nms(kpts_db: Sequence[dict], thresh: float, sigmas=None, in_vis_thre=None) -> List[int]:
    """``misc/nms/nms.py:97-122``: greedy NMS with OKS as the overlap; returns the indices to keep, best first."""
    if len(kpts_db) == 0:
        return []
    scores, kpts, areas = _db_arrays(kpts_db)
    n, joints = kpts.shape[0], kpts.shape[1] // 3
    sg = _sigmas(sigmas, joints)
    order = np.ascontiguousarray(scores.argsort()[::-1], np.int32)
    keep = np.zeros(n, np.int32)
    num = ctypes.c_int32(0)
    rc = _lib.load().hrn_oks_nms(keep.ctypes.data, ctypes.byref(num), kpts.ctypes.data, areas.ctypes.data, order.ctypes.data, n, joints,
                                 float(thresh), None if sg is None else sg.ctypes.data, _nan_if_none(in_vis_thre))
    if rc:
        raise ValueError("hrn_oks_nms: bad arguments")
    return [int(i) for i in keep[:num.value]]


def soft_oks_nms(kpts_db: Sequence[dict], thresh: float, sigmas=None, in_vis_thre=None):
    """``misc/nms/nms.py:138-180``: gaussian rescoring instead of removal, at most 20 kept; returns an index array."""
    if len(kpts_db) == 0:
        return []
    scores, kpts, areas = _db_arrays(kpts_db)
    n, joints = kpts.shape[0], kpts.shape[1] // 3
    sg = _sigmas(sigmas, joints)
    order64 = scores.argsort()[::-1]
    sorted_scores = np.ascontiguousarray(scores[order64], np.float64)
    order = np.ascontiguousarray(order64, np.int32)
    keep = np.zeros(max(n, 20), np.int32)
    num = ctypes.c_int32(0)
    rc = _lib.load().hrn_soft_oks_nms(keep.ctypes.data, ctypes.byref(num), kpts.ctypes.data, areas.ctypes.data, sorted_scores.ctypes.data,
                                      order.ctypes.data, n, joints, float(thresh), None if sg is None else sg.ctypes.data,
                                      _nan_if_none(in_vis_thre))
    if rc:
        raise ValueError("hrn_soft_oks_nms: bad arguments")
    return keep[:num.value].astype(np.intp)


def compute_similarity_matrices(bboxes_a, bboxes_b, poses_a, poses_b):
    """``misc/utils.py:372-384``: ``(box IoU, OKS)`` of every skeleton of a against every skeleton of b, float32."""
    assert len(bboxes_a) == len(poses_a) and len(bboxes_b) == len(poses_b)
    na, nb = len(poses_a), len(poses_b)
    pa = np.ascontiguousarray(poses_a, np.float32).reshape(na, -1, 3) if na else np.zeros((0, 1, 3), np.float32)
    pb = np.ascontiguousarray(poses_b, np.float32).reshape(nb, -1, 3) if nb else np.zeros((0, 1, 3), np.float32)
    joints = pa.shape[1] if na else pb.shape[1]
    if na and nb and pa.shape[1] != pb.shape[1]:
        raise ValueError("the two sets of skeletons have different joint counts")
    ba = np.ascontiguousarray(np.asarray(bboxes_a, np.float64).reshape(na, 4))
    bb = np.ascontiguousarray(np.asarray(bboxes_b, np.float64).reshape(nb, 4))
    result_bbox = np.zeros((na, nb), dtype=np.float32)
    result_pose = np.zeros((na, nb), dtype=np.float32)
    if na and nb:
        rc = _lib.load().hrn_pose_similarity(ba.ctypes.data, pa.ctypes.data, na, bb.ctypes.data, pb.ctypes.data, nb, joints,
                                             result_bbox.ctypes.data, result_pose.ctypes.data)
        if rc:
            raise ValueError("hrn_pose_similarity: bad arguments")
    return result_bbox, result_pose


def assignment(cost) -> List[tuple]:
    """What ``munkres.Munkres().compute(cost)`` returns (``misc/utils.py:406-407``): the pairs of a minimum-cost matching."""
    c = np.ascontiguousarray(cost, np.float64)
    if c.ndim != 2 or c.size == 0:
        return []
    row_to_col = np.full(c.shape[0], -1, np.int32)
    rc = _lib.load().hrn_assignment(c.ctypes.data, c.shape[0], c.shape[1], row_to_col.ctypes.data)
    if rc:
        raise ValueError("hrn_assignment: costs must be finite")
    return [(r, int(col)) for r, col in enumerate(row_to_col) if col >= 0]


def find_person_id_associations(boxes, pts, prev_boxes, prev_pts, prev_person_ids, next_person_id=0, pose_alpha=0.5,
                                similarity_threshold=0.5, smoothing_alpha=0.):
    """``misc/utils.py:387-429``: match the current skeletons to the previous frame's, carry the ids over, smooth matched
    boxes / joints in place, number the new people from ``next_person_id``.  Returns ``(boxes, pts, person_ids)``."""
    sim_box, sim_pose = compute_similarity_matrices(boxes, prev_boxes, pts, prev_pts)
    # the blend keeps the reference's operand order (float32 matrix * python float, pose term first): its rounding decides
    # which pairs clear the threshold
    similarity_matrix = sim_pose * pose_alpha + sim_box * (1 - pose_alpha)
    pairs = assignment((1 - similarity_matrix).tolist())
    person_ids = np.full(len(pts), -1, dtype=np.int32)

    def blend(now, before):   # linear temporal filter; assigning the result casts it to the caller's dtype (int32 boxes truncate)
        return (1 - smoothing_alpha) * now + smoothing_alpha * before

    for cur, prev in pairs:
        if not similarity_matrix[cur, prev] > similarity_threshold:
            continue
        person_ids[cur] = prev_person_ids[prev]
        if smoothing_alpha:
            boxes[cur], pts[cur] = blend(boxes[cur], prev_boxes[prev]), blend(pts[cur], prev_pts[prev])
    fresh = person_ids == -1
    person_ids[fresh] = np.arange(next_person_id, next_person_id + np.sum(fresh))
    return boxes, pts, person_ids


def inverse_affine(center, scale, pixel_std, output_size) -> np.ndarray:
    """The 2x3 matrix ``get_affine_transform(center, scale, pixel_std, 0, output_size, inv=1)`` returns
    (``misc/utils.py:44-76``): heat-map coordinates back to image coordinates for an unrotated crop.  The reference builds
    three float32 point pairs and lets ``cv2.getAffineTransform`` solve for the matrix in float64; the same pairs are built
    here with the same float32 roundings and the 6x6 system is solved by LU in float64 (``numpy.linalg.solve``)."""
    scale = np.asarray(scale)
    if scale.ndim == 0:
        scale = np.array([scale, scale])
    scale_tmp = scale * 1.0 * pixel_std
    src_w, dst_w, dst_h = scale_tmp[0], output_size[0], output_size[1]
    shift = np.array([0, 0], dtype=np.float32)
    src, dst = np.zeros((3, 2), dtype=np.float32), np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale_tmp * shift
    src[1, :] = center + [0 * 1.0 - (src_w * -0.5) * 0.0, 0 * 0.0 + (src_w * -0.5) * 1.0] + scale_tmp * shift   # get_dir, rot = 0
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + np.array([0, dst_w * -0.5], np.float32)
    for pts in (src, dst):                                   # get_3rd_point: b + (-(a - b).y, (a - b).x)
        direct = pts[0, :] - pts[1, :]
        pts[2, :] = pts[1, :] + np.array([-direct[1], direct[0]], dtype=np.float32)
    a = np.zeros((6, 6), np.float64)
    b = np.zeros(6, np.float64)
    for k in range(3):                                       # dst -> src (inv = 1)
        a[2 * k, 0:3] = (dst[k, 0], dst[k, 1], 1.0)
        a[2 * k + 1, 3:6] = (dst[k, 0], dst[k, 1], 1.0)
        b[2 * k], b[2 * k + 1] = src[k, 0], src[k, 1]
    return np.linalg.solve(a, b).reshape(2, 3)


def transform_preds(coords, center, scale, pixel_std, output_size) -> np.ndarray:
    """``misc/utils.py:116-123``: (J, 2) heat-map coordinates of one crop -> image coordinates, float32.  With the
    ``preds`` of ``NativeHRNet.predict_flip_tta`` this completes ``get_final_preds`` (``misc/utils.py:154-180``)."""
    coords = np.asarray(coords.detach().cpu().numpy() if hasattr(coords, "detach") else coords)
    target = np.zeros(coords.shape, dtype=np.float32)
    trans = inverse_affine(center, scale, pixel_std, output_size)
    for p in range(coords.shape[0]):
        target[p, 0:2] = np.dot(trans, np.array([coords[p, 0], coords[p, 1], 1.]).T)[:2]
    return target


def final_preds(preds, center, scale, pixel_std, heatmap_size) -> np.ndarray:
    """the "Transform back" loop of ``get_final_preds`` (``misc/utils.py:176-178``) over a batch: preds (n, J, 2) in heat-map
    pixels (x, y), center / scale (n, 2) as the dataset provides them, heatmap_size = (width, height)."""
    preds = np.asarray(preds.detach().cpu().numpy() if hasattr(preds, "detach") else preds)
    out = np.empty(preds.shape, np.float32)
    for i in range(preds.shape[0]):
        out[i] = transform_preds(preds[i], center[i], scale[i], pixel_std, heatmap_size)
    return out
