"""Host-side pose post-processing (second half): OKS NMS and the tracker's association step on the
native library against the outputs of the reference's own functions (tests/golden/make_tracking_golden.py ran the
unmodified misc/nms/nms.py and misc/utils.py).  Index results exact; similarity matrices to the last float32 bit."""
import numpy as np
import pytest

from conftest import golden, load_pkg

G = golden("tracking_cases")
pp = load_pkg("postproc")


@pytest.mark.parametrize("k", [int(v) for v in G["oks_cases"]])
def test_oks_nms_equals_reference(k):
    kpts, scores, areas = G["oks%d_kpts" % k], G["oks%d_scores" % k], G["oks%d_areas" % k]
    thresh, vis = float(G["oks%d_thresh" % k]), float(G["oks%d_vis" % k])
    vis = None if np.isnan(vis) else vis
    db = [{"keypoints": kpts[i], "score": scores[i], "area": areas[i]} for i in range(len(kpts))]
    assert pp.oks_nms(db, thresh, None, vis) == G["oks%d_keep" % k].tolist()
    np.testing.assert_array_equal(pp.soft_oks_nms(db, thresh, None, vis), G["oks%d_soft_keep" % k])
    # explicit sigmas == the default ones
    sig = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0
    assert pp.oks_nms(db, thresh, sig, vis) == G["oks%d_keep" % k].tolist()
    assert pp.oks_nms([], thresh) == [] and len(pp.soft_oks_nms([], thresh)) == 0


@pytest.mark.parametrize("k", [int(v) for v in G["track_cases"]])
def test_tracker_equals_reference(k):
    g = {n: G["track%d_%s" % (k, n)] for n in ("boxes", "pts", "prev_boxes", "prev_pts", "prev_ids", "next_id", "params", "sim_bbox",
                                               "sim_pose", "out_boxes", "out_pts", "out_ids")}
    sim_bbox, sim_pose = pp.compute_similarity_matrices(g["boxes"], g["prev_boxes"], g["pts"], g["prev_pts"])
    assert sim_bbox.dtype == np.float32 and sim_pose.dtype == np.float32
    np.testing.assert_array_equal(sim_bbox, g["sim_bbox"])
    np.testing.assert_array_equal(sim_pose, g["sim_pose"])
    alpha, thr, smooth = (float(v) for v in g["params"])
    boxes, pts = g["boxes"].copy(), g["pts"].copy()
    b, p, ids = pp.find_person_id_associations(boxes, pts, g["prev_boxes"], g["prev_pts"], g["prev_ids"], next_person_id=int(g["next_id"]),
                                               pose_alpha=alpha, similarity_threshold=thr, smoothing_alpha=smooth)
    assert b is boxes and p is pts and ids.dtype == np.int32          # smoothed in place, like the reference
    np.testing.assert_array_equal(ids, g["out_ids"])
    np.testing.assert_array_equal(b, g["out_boxes"])
    np.testing.assert_array_equal(p, g["out_pts"])


def test_non_coco_joint_count_uses_the_float32_sigmas():
    """16 joints (MPII): sigmas are float32 ones / 10 in the reference (misc/utils.py:346-347), so the first division is a
    float32 one -- restated here in numpy and compared bit for bit."""
    rng = np.random.default_rng(3)
    pa = rng.uniform(0, 200, (3, 16, 3)).astype(np.float32)
    pb = (pa[[2, 0]] + rng.normal(0, 3, (2, 16, 3))).astype(np.float32)
    ba = np.asarray([[0, 0, 100, 200], [50, 20, 90, 180], [10, 10, 60, 90]], np.int32)
    bb = ba[[2, 0]] + 2
    _, pose = pp.compute_similarity_matrices(ba, bb, pa, pb)
    sig = np.ones((16,), dtype=np.float32) / 10.0
    var = (sig * 2) ** 2
    area = lambda b: (b[2] - b[0]) * (b[3] - b[1])
    ref = np.zeros((3, 2), np.float32)
    for i in range(3):
        for j in range(2):
            dx, dy = pb[j, :, 1] - pa[i, :, 1], pb[j, :, 0] - pa[i, :, 0]
            e = (dx ** 2 + dy ** 2) / var / ((area(ba[i]) + area(bb[j])) / 2 + np.spacing(1)) / 2
            e = e[e <= 29]
            ref[i, j] = np.sum(np.exp(-e)) / e.shape[0] if e.shape[0] else 0.0
    np.testing.assert_array_equal(pose, ref)


def test_assignment_is_optimal_and_shaped_like_munkres():
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(11)
    for rows, cols in [(1, 1), (3, 3), (2, 5), (6, 2), (7, 7), (12, 9)]:
        c = rng.uniform(0, 1, (rows, cols))
        pairs = pp.assignment(c)
        assert len(pairs) == min(rows, cols) and len({r for r, _ in pairs}) == len(pairs) == len({q for _, q in pairs})
        r, q = linear_sum_assignment(c)
        assert abs(sum(c[a, b] for a, b in pairs) - c[r, q].sum()) < 1e-12
        assert sorted(pairs) == sorted(zip(r.tolist(), q.tolist()))       # unique optimum on continuous random costs
    assert pp.assignment(np.zeros((0, 3))) == [] and pp.assignment(np.zeros((2, 0))) == []
    with pytest.raises(ValueError):
        pp.assignment(np.full((2, 2), np.nan))


def test_empty_sides():
    b, p = pp.compute_similarity_matrices(np.zeros((0, 4), np.int32), np.zeros((2, 4), np.int32), np.zeros((0, 17, 3), np.float32),
                                          np.zeros((2, 17, 3), np.float32))
    assert b.shape == (0, 2) and p.shape == (0, 2)


def test_inverse_affine_of_the_evaluation_decode_equals_reference():
    """transform_preds (misc/utils.py:116-123): equal to the reference's function on the fixture, and to the closed form an
    unrotated crop has (uniform scale src_w / dst_w about the centre) within a float32 ulp of the coordinate."""
    for k in range(len(G["affine_out"])):
        center, scale, coords = G["affine_center"][k], G["affine_scale"][k], G["affine_coords"][k]
        w, h = (int(v) for v in G["affine_size"][k])
        got = pp.transform_preds(coords, center, scale, 200, [w, h])
        assert got.dtype == np.float32
        np.testing.assert_array_equal(got, G["affine_out"][k])
        s = float(scale[0]) * 200 / w
        closed = np.stack([(coords[:, 0].astype(np.float64) - w * 0.5) * s + float(center[0]),
                           (coords[:, 1].astype(np.float64) - h * 0.5) * s + float(center[1])], 1)
        assert np.abs(closed - got).max() < 2e-4
    batch = pp.final_preds(G["affine_coords"][[1, 3]], G["affine_center"][[1, 3]], G["affine_scale"][[1, 3]], 200, [72, 96])
    np.testing.assert_array_equal(batch, G["affine_out"][[1, 3]])
