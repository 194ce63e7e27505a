// Host-side pose post-processing of the reference, behind the same C ABI (include/hrnet_mi355.h):
//   * OKS non-maximum suppression, hard and soft      misc/nms/nms.py:75-180   (evaluation: datasets/COCO.py:371-374)
//   * tracker: box-IoU / OKS similarity matrices and the optimal assignment   misc/utils.py:251-429
//     (live demo: scripts/live-demo.py:120-123)
// These are O(people^2 * joints) on a handful of skeletons: they are host code in the reference (numpy + the munkres
// package) and stay host code here -- a kernel launch costs more than the whole computation.  What matters is that the
// numbers are the reference's: float64 arithmetic in numpy's operation order (its pairwise summation included), float32
// where the reference's arrays are float32, and the reference's quirks kept (the visibility mask that only looks at the
// candidate, `e <= 2^32 - 1` read by Python as `e <= 29`).  No fused multiply-adds: numpy has none.
#include "../../include/hrnet_mi355.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>

#pragma clang fp contract(off)

namespace {

const double kCocoSigmas[17] = {.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89};
const double kSpacing1 = 2.220446049250313e-16;  // np.spacing(1)

// numpy's pairwise summation for n <= 128 contiguous doubles (numpy/core/src/umath/loops_utils.h.src: pairwise_sum)
double np_sum(const double *a, int n) {
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_sum(a, n2) + np_sum(a + n2, n - n2);
}

// misc/nms/nms.py:75-94: OKS of skeleton g against skeleton d, both flat (x, y, v) * J, float64
double oks_flat(const double *g, const double *d, double a_g, double a_d, const double *vars, int J, bool use_vis, double vis,
                std::vector<double> &tmp) {
    tmp.clear();
    const double denom = (a_g + a_d) / 2 + kSpacing1;
    for (int j = 0; j < J; ++j) {
        const double dx = d[3 * j] - g[3 * j], dy = d[3 * j + 1] - g[3 * j + 1];
        const double e = (dx * dx + dy * dy) / vars[j] / denom / 2;
        // `list(vg > t) and list(vd > t)` is the SECOND list whenever the first is non-empty: the mask is the candidate's
        if (use_vis && !(d[3 * j + 2] > vis)) continue;
        tmp.push_back(std::exp(-e));
    }
    return tmp.empty() ? 0.0 : np_sum(tmp.data(), (int)tmp.size()) / (double)tmp.size();
}

void make_vars(const double *sigmas, int J, std::vector<double> &vars) {
    vars.resize(J);
    for (int j = 0; j < J; ++j) {
        const double s = sigmas ? sigmas[j] : kCocoSigmas[j] / 10.0;
        vars[j] = (s * 2) * (s * 2);
    }
}

}  // namespace

extern "C" {

int hrn_oks_nms(int32_t *keep_out, int32_t *num_out, const double *kpts, const double *areas, const int32_t *order, int n, int J,
                double thresh, const double *sigmas, double in_vis_thre) {
    if (!keep_out || !num_out || n < 0 || J <= 0 || (n && (!kpts || !areas || !order)) || (!sigmas && J != 17)) return 1;
    std::vector<double> vars, tmp;
    make_vars(sigmas, J, vars);
    const bool use_vis = !std::isnan(in_vis_thre);
    std::vector<int> cur(order, order + n), nxt;
    int kept = 0;
    while (!cur.empty()) {
        const int i = cur[0];
        keep_out[kept++] = i;
        nxt.clear();
        for (size_t k = 1; k < cur.size(); ++k) {
            const int c = cur[k];
            const double o = oks_flat(kpts + (size_t)i * 3 * J, kpts + (size_t)c * 3 * J, areas[i], areas[c], vars.data(), J, use_vis,
                                      in_vis_thre, tmp);
            if (o <= thresh) nxt.push_back(c);
        }
        cur.swap(nxt);
    }
    *num_out = kept;
    return 0;
}

int hrn_soft_oks_nms(int32_t *keep_out, int32_t *num_out, const double *kpts, const double *areas, const double *scores_sorted,
                     const int32_t *order, int n, int J, double thresh, const double *sigmas, double in_vis_thre) {
    if (!keep_out || !num_out || n < 0 || J <= 0 || (n && (!kpts || !areas || !order || !scores_sorted)) || (!sigmas && J != 17))
        return 1;
    std::vector<double> vars, tmp;
    make_vars(sigmas, J, vars);
    const bool use_vis = !std::isnan(in_vis_thre);
    std::vector<int> cur(order, order + n);
    std::vector<double> sc(scores_sorted, scores_sorted + n);
    const int max_dets = 20;  // misc/nms/nms.py:156
    int kept = 0;
    while (!cur.empty() && kept < max_dets) {
        const int i = cur[0];
        const size_t m = cur.size() - 1;
        std::vector<int> rest(cur.begin() + 1, cur.end());
        std::vector<double> rs(m);
        for (size_t k = 0; k < m; ++k) {
            const int c = rest[k];
            const double o = oks_flat(kpts + (size_t)i * 3 * J, kpts + (size_t)c * 3 * J, areas[i], areas[c], vars.data(), J, use_vis,
                                      in_vis_thre, tmp);
            rs[k] = sc[k + 1] * std::exp(-(o * o) / thresh);  // rescore(..., type='gaussian')
        }
        std::vector<int> idx(m);
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return rs[a] > rs[b]; });  // argsort()[::-1]
        cur.resize(m), sc.resize(m);
        for (size_t k = 0; k < m; ++k) cur[k] = rest[idx[k]], sc[k] = rs[idx[k]];
        keep_out[kept++] = i;
    }
    *num_out = kept;
    return 0;
}

// misc/utils.py:372-384 (compute_similarity_matrices) with bbox_iou (:318-334) and oks_iou (:341-369)
int hrn_pose_similarity(const double *boxes_a, const float *poses_a, int na, const double *boxes_b, const float *poses_b, int nb, int J,
                        float *sim_bbox, float *sim_pose) {
    if (na < 0 || nb < 0 || J <= 0 || (na && (!boxes_a || !poses_a)) || (nb && (!boxes_b || !poses_b)) ||
        (na && nb && (!sim_bbox || !sim_pose)))
        return 1;
    auto area = [](const double *b) { return (b[2] - b[0]) * (b[3] - b[1]); };
    // sigmas: COCO's as float64 for 17 joints, float32 ones / 10 otherwise (:343-348) -- the dtype decides where the
    // first division is rounded
    const bool coco = J == 17;
    std::vector<double> vars64(J);
    std::vector<float> vars32(J);
    for (int j = 0; j < J; ++j) {
        const double s = kCocoSigmas[j % 17] / 10.0;
        vars64[j] = (s * 2) * (s * 2);
        const float s32 = 1.0f / 10.0f;
        vars32[j] = (s32 * 2) * (s32 * 2);
    }
    std::vector<double> tmp;
    for (int i = 0; i < na; ++i) {
        const float *g = poses_a + (size_t)i * J * 3;
        const double a_g = area(boxes_a + 4 * i);
        for (int k = 0; k < nb; ++k) {
            const float *d = poses_b + (size_t)k * J * 3;
            const double denom = (a_g + area(boxes_b + 4 * k)) / 2 + kSpacing1;
            tmp.clear();
            for (int j = 0; j < J; ++j) {
                const float dy = d[3 * j] - g[3 * j], dx = d[3 * j + 1] - g[3 * j + 1];   // (y, x, v) float32 arrays
                const float sq = dx * dx + dy * dy;
                const double e = (coco ? (double)sq / vars64[j] : (double)(sq / vars32[j])) / denom / 2;
                if (e <= 29) tmp.push_back(std::exp(-e));   // `e[e <= 2^32 - 1]`: ^ is XOR in Python, 2 ^ 31 == 29
            }
            sim_pose[(size_t)i * nb + k] = (float)(tmp.empty() ? 0.0 : np_sum(tmp.data(), (int)tmp.size()) / (double)tmp.size());
            // box IoU: intersection limits, zero area when disjoint, union = a + b - i (:269-334)
            const double *p = boxes_a + 4 * i, *q = boxes_b + 4 * k;
            const double x1 = std::max(p[0], q[0]), x2 = std::min(p[2], q[2]), y1 = std::max(p[1], q[1]), y2 = std::min(p[3], q[3]);
            const double area_i = (x2 < x1 || y2 < y1) ? 0.0 : (x2 - x1) * (y2 - y1);
            const double area_u = area(p) + area(q) - area_i;
            sim_bbox[(size_t)i * nb + k] = (float)(area_i / area_u);
        }
    }
    return 0;
}

// Minimum-cost assignment of a rows x cols matrix (what Munkres().compute() returns, misc/utils.py:406-407): every row gets
// a column when rows <= cols, otherwise every column gets a row; row_to_col[r] = column or -1.  Shortest augmenting paths
// with potentials, O(n^2 m).
int hrn_assignment(const double *cost, int rows, int cols, int32_t *row_to_col) {
    if (rows < 0 || cols < 0 || (rows && !row_to_col) || (rows && cols && !cost)) return 1;
    for (int r = 0; r < rows; ++r) row_to_col[r] = -1;
    if (rows == 0 || cols == 0) return 0;
    const bool transposed = rows > cols;
    const int n = transposed ? cols : rows, m = transposed ? rows : cols;  // n <= m
    auto c = [&](int i, int j) { return transposed ? cost[(size_t)j * cols + i] : cost[(size_t)i * cols + j]; };
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<double> u(n + 1, 0.0), v(m + 1, 0.0), minv(m + 1);
    std::vector<int> p(m + 1, 0), way(m + 1, 0);
    std::vector<char> used(m + 1);
    for (int i = 1; i <= n; ++i) {
        p[0] = i;
        int j0 = 0;
        std::fill(minv.begin(), minv.end(), inf);
        std::fill(used.begin(), used.end(), 0);
        do {
            used[j0] = 1;
            const int i0 = p[j0];
            double delta = inf;
            int j1 = 0;
            for (int j = 1; j <= m; ++j)
                if (!used[j]) {
                    const double cur = c(i0 - 1, j - 1) - u[i0] - v[j];
                    if (cur < minv[j]) minv[j] = cur, way[j] = j0;
                    if (minv[j] < delta) delta = minv[j], j1 = j;
                }
            if (j1 == 0) return 2;  // NaN / inf costs: no augmenting path
            for (int j = 0; j <= m; ++j)
                if (used[j])
                    u[p[j]] += delta, v[j] -= delta;
                else
                    minv[j] -= delta;
            j0 = j1;
        } while (p[j0] != 0);
        do {
            const int j1 = way[j0];
            p[j0] = p[j1];
            j0 = j1;
        } while (j0);
    }
    for (int j = 1; j <= m; ++j)
        if (p[j]) {
            const int small = p[j] - 1, big = j - 1;
            if (transposed)
                row_to_col[big] = small;
            else
                row_to_col[small] = big;
        }
    return 0;
}

}  // extern "C"
