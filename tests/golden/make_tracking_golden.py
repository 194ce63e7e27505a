"""Golden vectors for the pose post-processing the reference does on the host (second half):
``oks_nms`` / ``soft_oks_nms`` (misc/nms/nms.py:75-180) and the tracker ``find_person_id_associations`` with its
similarity matrices (misc/utils.py:341-429), produced by IMPORTING THE UNMODIFIED REFERENCE FUNCTIONS in the build
container.  Stand-ins only for imports that are absent here and are not the code under test: ``cv2`` (unused by these
functions except ``getAffineTransform``, for which the stand-in solves the same three-point system in float64), the compiled
``cpu_nms`` / ``gpu_nms`` extensions (unused by the OKS functions) and ``munkres`` -- the
reference calls ``Munkres().compute(cost)`` for an optimal assignment; the stand-in returns scipy's
``linear_sum_assignment`` of the same matrix (the optimum is unique on these inputs: no tied costs).

    python tests/golden/make_tracking_golden.py      ->  tests/golden/tracking_cases.npz
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def install_stubs():
    from scipy.optimize import linear_sum_assignment

    class Munkres:
        def compute(self, cost):
            c = np.asarray(cost, dtype=np.float64)
            if c.size == 0:
                return []
            rows, cols = linear_sum_assignment(c)
            return list(zip(rows.tolist(), cols.tolist()))

    m = types.ModuleType("munkres")
    m.Munkres = Munkres
    sys.modules["munkres"] = m
    cv2 = types.ModuleType("cv2")

    def get_affine_transform(src, dst):
        """stand-in for cv2.getAffineTransform: the 2x3 matrix mapping three points onto three points, float64 LU solve"""
        a, b = np.zeros((6, 6)), np.zeros(6)
        for k in range(3):
            a[2 * k, 0:3] = (src[k, 0], src[k, 1], 1.0)
            a[2 * k + 1, 3:6] = (src[k, 0], src[k, 1], 1.0)
            b[2 * k], b[2 * k + 1] = dst[k, 0], dst[k, 1]
        return np.linalg.solve(a, b).reshape(2, 3)

    cv2.getAffineTransform = get_affine_transform
    sys.modules["cv2"] = cv2
    for name in ("cpu_nms", "gpu_nms"):
        mod = types.ModuleType(name)
        setattr(mod, name, lambda *a, **k: (_ for _ in ()).throw(RuntimeError("not under test")))
        sys.modules[name] = mod


def people(rng, n, j, frame=(480, 640), jitter=0.0, base=None):
    """n synthetic skeletons (n, j, 3) float32 (y, x, confidence) with their int32 boxes (x1, y1, x2, y2)"""
    if base is None:
        cy, cx = rng.uniform(80, frame[0] - 80, n), rng.uniform(80, frame[1] - 80, n)
        sz = rng.uniform(40, 160, n)
        pts = np.empty((n, j, 3), np.float32)
        pts[:, :, 0] = cy[:, None] + rng.normal(0, 1, (n, j)) * sz[:, None] * 0.4
        pts[:, :, 1] = cx[:, None] + rng.normal(0, 1, (n, j)) * sz[:, None] * 0.25
        pts[:, :, 2] = rng.uniform(0.05, 1.0, (n, j))
    else:
        pts = base.copy()
        pts[:, :, :2] += rng.normal(0, jitter, pts[:, :, :2].shape).astype(np.float32)
        pts[:, :, 2] = np.clip(pts[:, :, 2] + rng.normal(0, 0.05, pts.shape[:2]), 0.01, 1).astype(np.float32)
    boxes = np.stack([pts[:, :, 1].min(1) - 5, pts[:, :, 0].min(1) - 5, pts[:, :, 1].max(1) + 5, pts[:, :, 0].max(1) + 5], 1)
    return pts, np.round(boxes).astype(np.int32)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "misc", "nms"))
    from misc import utils as U
    import nms as N                                     # misc/nms/nms.py (imports cpu_nms / gpu_nms by bare name)

    rng = np.random.default_rng(7)
    out = {}
    # ---- tracker: current vs previous frame (misc/utils.py:372-429), as scripts/live-demo.py:120-123 calls it
    cases = []
    for k, (n_prev, n_cur, j, jitter, params) in enumerate([
            (4, 4, 17, 3.0, dict(pose_alpha=0.2, similarity_threshold=0.4, smoothing_alpha=0.1)),
            (5, 3, 17, 6.0, dict(pose_alpha=0.5, similarity_threshold=0.5, smoothing_alpha=0.0)),
            (2, 6, 17, 2.0, dict(pose_alpha=0.2, similarity_threshold=0.4, smoothing_alpha=0.1)),
            (3, 3, 16, 40.0, dict(pose_alpha=0.8, similarity_threshold=0.3, smoothing_alpha=0.5)),
            (1, 1, 17, 1.0, dict(pose_alpha=0.5, similarity_threshold=0.5, smoothing_alpha=0.25)),
            (6, 6, 17, 300.0, dict(pose_alpha=0.5, similarity_threshold=0.5, smoothing_alpha=0.1))]):
        prev_pts, prev_boxes = people(rng, n_prev, j)
        common = min(n_prev, n_cur)
        perm = rng.permutation(n_prev)[:common]
        cur_pts, cur_boxes = people(rng, common, j, jitter=jitter, base=prev_pts[perm])
        if n_cur > common:
            extra_pts, extra_boxes = people(rng, n_cur - common, j)
            cur_pts, cur_boxes = np.concatenate([cur_pts, extra_pts]), np.concatenate([cur_boxes, extra_boxes])
        order = rng.permutation(n_cur)
        cur_pts, cur_boxes = cur_pts[order], cur_boxes[order]
        prev_ids = (rng.permutation(20)[:n_prev]).astype(np.int32)
        next_id = int(prev_ids.max()) + 1
        sim_bbox, sim_pose = U.compute_similarity_matrices(cur_boxes, prev_boxes, cur_pts, prev_pts)
        b, p, ids = U.find_person_id_associations(cur_boxes.copy(), cur_pts.copy(), prev_boxes, prev_pts, prev_ids,
                                                  next_person_id=next_id, **params)
        for name, v in (("boxes", cur_boxes), ("pts", cur_pts), ("prev_boxes", prev_boxes), ("prev_pts", prev_pts),
                        ("prev_ids", prev_ids), ("next_id", np.int32(next_id)),
                        ("params", np.asarray([params["pose_alpha"], params["similarity_threshold"], params["smoothing_alpha"]])),
                        ("sim_bbox", sim_bbox), ("sim_pose", sim_pose), ("out_boxes", b), ("out_pts", p), ("out_ids", ids)):
            out["track%d_%s" % (k, name)] = v
        cases.append(k)
    out["track_cases"] = np.asarray(cases, np.int32)
    # ---- OKS NMS (misc/nms/nms.py:97-180), as datasets/COCO.py:371-374 calls it: per image, list of dicts
    nms_cases = []
    for k, (n, j, thresh, vis) in enumerate([(12, 17, 0.9, None), (12, 17, 0.5, None), (9, 17, 0.9, 0.2), (25, 17, 0.7, None),
                                              (1, 17, 0.9, None), (7, 17, 0.3, 0.5)]):
        base_pts, _ = people(rng, max(1, n // 3), j)
        idx = rng.integers(0, len(base_pts), n)
        pts, boxes = people(rng, n, j, jitter=4.0, base=base_pts[idx])
        kpts = np.stack([pts[:, :, 1], pts[:, :, 0], pts[:, :, 2]], 2).astype(np.float64)     # COCO order: x, y, score
        scores = rng.uniform(0.1, 1.0, n)
        areas = ((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])).astype(np.float64)
        db = [{"keypoints": kpts[i], "score": scores[i], "area": areas[i]} for i in range(n)]
        keep = np.asarray(N.oks_nms(db, thresh, None, vis), np.int32)
        soft = np.asarray(N.soft_oks_nms(db, thresh, None, vis), np.int32)
        first = N.oks_iou(kpts[0].flatten(), np.stack([q.flatten() for q in kpts]), areas[0], areas, None, vis)
        for name, v in (("kpts", kpts), ("scores", scores), ("areas", areas), ("thresh", np.float64(thresh)),
                        ("vis", np.float64(np.nan if vis is None else vis)), ("keep", keep), ("soft_keep", soft), ("oks_row0", first)):
            out["oks%d_%s" % (k, name)] = v
        nms_cases.append(k)
    out["oks_cases"] = np.asarray(nms_cases, np.int32)
    # ---- inverse affine of the evaluation decode (misc/utils.py:116-123, 176-178): transform_preds per crop
    import torch
    centers, scales, coords, sizes, results = [], [], [], [], []
    for k in range(12):
        w, h = (72, 96) if k % 2 else (48, 64)
        center = rng.uniform(50, 900, 2).astype(np.float32)
        scale = (np.array([rng.uniform(40, 400) / 200, rng.uniform(40, 600) / 200], dtype=np.float32) * 1.25).astype(np.float32)
        c = np.stack([rng.uniform(0, w, 17), rng.uniform(0, h, 17)], 1).astype(np.float32)
        centers.append(center), scales.append(scale), coords.append(c), sizes.append((w, h))
        results.append(U.transform_preds(torch.from_numpy(c), center, scale, 200, [w, h]).numpy())
    out.update(affine_center=np.stack(centers), affine_scale=np.stack(scales), affine_coords=np.stack(coords),
               affine_size=np.asarray(sizes, np.int32), affine_out=np.stack(results))
    path = os.path.join(HERE, "tracking_cases.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB): %d tracker cases, %d OKS-NMS cases" % (path, os.path.getsize(path) / 1024, len(cases), len(nms_cases)))


if __name__ == "__main__":
    main()
