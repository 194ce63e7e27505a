// Greedy IoU non-maximum suppression for score-sorted boxes -- the one native component of the reference
// (misc/nms/nms_kernel.cu, C interface misc/nms/gpu_nms.hpp `_nms`, Cython caller gpu_nms.pyx:19-34), rebuilt for
// gfx950.  Same contract: boxes (n, dim >= 5) float32 sorted by score descending, [x1, y1, x2, y2, score], pixel
// convention (+1), a box is dropped when its IoU with an earlier kept box is > thresh; returns the kept indices
// (into the sorted array) in order.
//
//   nms_mask_kernel   one wave per (64-row, 64-column) tile of the upper triangle: lane = row box in registers, the
//                     column boxes come in by wave-uniform (scalar) loads; 64 IoUs -> one 64-bit mask word per lane.
//                     No LDS: a wave IS the 64-box tile.
//   nms_sweep_kernel  the greedy pass, also on the GPU (the reference copies the n x n/64 mask to the host and loops
//                     there): one wave, lane l owns suppression word l (n <= 4096) -- per box one ballot-free bit
//                     test and, if kept, one coalesced 512-byte OR of its mask row.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/hrnet_mi355.h"

namespace {

#pragma clang fp contract(off)
__device__ __forceinline__ float iou_px(const float a0, const float a1, const float a2, const float a3, const float *b) {
    const float left = fmaxf(a0, b[0]), right = fminf(a2, b[2]);
    const float top = fmaxf(a1, b[1]), bottom = fminf(a3, b[3]);
    const float width = fmaxf(right - left + 1.f, 0.f), height = fmaxf(bottom - top + 1.f, 0.f);
    const float inter = width * height;
    const float sa = (a2 - a0 + 1.f) * (a3 - a1 + 1.f);
    const float sb = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
    return inter / (sa + sb - inter);
}

__global__ __launch_bounds__(64) void nms_mask_kernel(const float *boxes, int n, int dim, float thresh,
                                                      unsigned long long *mask, int col_blocks) {
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb) return;  // lower triangle: never read (a box only suppresses later ones)
    const int row = rb * 64 + threadIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (row < n) a0 = boxes[(size_t)row * dim], a1 = boxes[(size_t)row * dim + 1], a2 = boxes[(size_t)row * dim + 2], a3 = boxes[(size_t)row * dim + 3];
    const int ncol = min(64, n - cb * 64);
    unsigned long long t = 0;
    for (int i = 0; i < ncol; ++i) {
        const int col = cb * 64 + i;           // wave-uniform: the four loads below are scalar loads
        if (row < n && col > row && iou_px(a0, a1, a2, a3, boxes + (size_t)col * dim) > thresh) t |= 1ull << i;
    }
    if (row < n) mask[(size_t)row * col_blocks + cb] = t;
}

__global__ __launch_bounds__(64) void nms_sweep_kernel(const unsigned long long *mask, int n, int col_blocks, int *keep, int *num_out) {
    const int lane = threadIdx.x;
    unsigned long long remv = 0;  // lane l: boxes 64l .. 64l+63 already suppressed
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned long long w = __shfl(remv, i >> 6);   // the owner lane's word, broadcast
        if (!((w >> (i & 63)) & 1ull)) {                     // wave-uniform
            if (lane == 0) keep[kept] = i;
            ++kept;
            if (lane < col_blocks && lane >= (i >> 6)) remv |= mask[(size_t)i * col_blocks + lane];
        }
    }
    if (lane == 0) *num_out = kept;
}

thread_local std::string g_nms_error;

bool ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    g_nms_error = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

}  // namespace

extern "C" const char *hrn_nms_last_error(void) { return g_nms_error.c_str(); }

extern "C" int hrn_nms(int32_t *keep_out, int32_t *num_out, const float *boxes_host, int boxes_num, int boxes_dim,
                       float nms_overlap_thresh, int device_id) {
    if (!keep_out || !num_out || boxes_num < 0 || boxes_dim < 5 || (boxes_num > 0 && !boxes_host)) {
        g_nms_error = "bad arguments";
        return 1;
    }
    *num_out = 0;
    if (boxes_num == 0) return 0;
    if (boxes_num > 4096) {
        g_nms_error = "at most 4096 boxes (one suppression word per lane of the sweeping wave)";
        return 2;
    }
    if (!ok(hipSetDevice(device_id), "hipSetDevice")) return 3;
    const int col_blocks = (boxes_num + 63) / 64;
    float *boxes_dev = nullptr;
    unsigned long long *mask_dev = nullptr;
    int *keep_dev = nullptr;
    const size_t bytes = (size_t)boxes_num * boxes_dim * sizeof(float);
    int rc = 0;
    if (!ok(hipMalloc((void **)&boxes_dev, bytes), "hipMalloc(boxes)") ||
        !ok(hipMalloc((void **)&mask_dev, (size_t)boxes_num * col_blocks * 8), "hipMalloc(mask)") ||
        !ok(hipMalloc((void **)&keep_dev, ((size_t)boxes_num + 1) * 4), "hipMalloc(keep)") ||
        !ok(hipMemcpy(boxes_dev, boxes_host, bytes, hipMemcpyHostToDevice), "hipMemcpy(boxes)")) {
        rc = 3;
    } else {
        hipLaunchKernelGGL(nms_mask_kernel, dim3(col_blocks, col_blocks), dim3(64), 0, 0, boxes_dev, boxes_num, boxes_dim,
                           nms_overlap_thresh, mask_dev, col_blocks);
        hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(64), 0, 0, mask_dev, boxes_num, col_blocks, keep_dev, keep_dev + boxes_num);
        if (!ok(hipGetLastError(), "nms launch") ||
            !ok(hipMemcpy(num_out, keep_dev + boxes_num, 4, hipMemcpyDeviceToHost), "hipMemcpy(num_out)") ||
            !ok(hipMemcpy(keep_out, keep_dev, (size_t)(*num_out) * 4, hipMemcpyDeviceToHost), "hipMemcpy(keep)"))
            rc = 4;
    }
    if (boxes_dev) (void)hipFree(boxes_dev);
    if (mask_dev) (void)hipFree(mask_dev);
    if (keep_dev) (void)hipFree(keep_dev);
    return rc;
}
