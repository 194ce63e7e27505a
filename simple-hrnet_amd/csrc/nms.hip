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
//   nms_sweep_lds_kernel  the same pass for n > 4096 (the reference has no cap): the suppression words live in LDS
//                     (8 bytes per 64 boxes), one wave, no barrier needed -- a wave's LDS operations execute in order.
// Device scratch is kept per device between calls (grown on demand), so a call costs two copies and two launches.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
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

__global__ __launch_bounds__(64) void nms_sweep_lds_kernel(const unsigned long long *mask, int n, int col_blocks, int *keep, int *num_out) {
    extern __shared__ unsigned long long remv_lds[];
    volatile unsigned long long *remv = remv_lds;
    const int lane = threadIdx.x;
    for (int c = lane; c < col_blocks; c += 64) remv[c] = 0;
    __builtin_amdgcn_wave_barrier();
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned long long w = remv[i >> 6];             // same address in every lane: a broadcast read
        if (!((w >> (i & 63)) & 1ull)) {                       // wave-uniform
            if (lane == 0) keep[kept] = i;
            ++kept;
            for (int c = (i >> 6) + lane; c < col_blocks; c += 64) remv[c] |= mask[(size_t)i * col_blocks + c];
            __builtin_amdgcn_wave_barrier();
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

// per-device scratch, reused across calls (the reference allocates and frees per call: nms_kernel.cu:120-143)
struct Scratch {
    std::mutex mu;
    float *boxes = nullptr;
    unsigned long long *mask = nullptr;
    int *keep = nullptr;
    size_t boxes_bytes = 0, mask_bytes = 0, keep_bytes = 0;
};
constexpr int kMaxDevices = 64;
Scratch g_scratch[kMaxDevices];

template <class T>
bool grow(T *&ptr, size_t &have, size_t need, const char *what) {
    if (need <= have) return true;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr, have = 0;
    const size_t cap = need + need / 2;
    if (!ok(hipMalloc((void **)&ptr, cap), what)) return false;
    have = cap;
    return true;
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
    const int col_blocks = (boxes_num + 63) / 64;
    // the n x ceil(n / 64) bit mask is the scratch that grows quadratically: 65536 boxes = 512 MiB, the cap (a detector
    // emits thousands; the reference itself allocates the same mask per call)
    if (boxes_num > 65536) {
        g_nms_error = "more than 65536 boxes (the n x n / 64 suppression mask would exceed 512 MiB)";
        return 2;
    }
    if (device_id < 0 || device_id >= kMaxDevices) {
        g_nms_error = "bad device id";
        return 1;
    }
    // the caller's current device is restored on every path out (the reference's _nms leaves it changed: gpu_nms.hpp)
    struct DeviceGuard {
        int prev = -1;
        DeviceGuard() { (void)hipGetDevice(&prev); }
        ~DeviceGuard() {
            if (prev >= 0) (void)hipSetDevice(prev);
        }
    } guard;
    if (!ok(hipSetDevice(device_id), "hipSetDevice")) return 3;
    Scratch &sc = g_scratch[device_id];
    std::lock_guard<std::mutex> lock(sc.mu);
    const size_t bytes = (size_t)boxes_num * boxes_dim * sizeof(float);
    if (!grow(sc.boxes, sc.boxes_bytes, bytes, "hipMalloc(boxes)") ||
        !grow(sc.mask, sc.mask_bytes, (size_t)boxes_num * col_blocks * 8, "hipMalloc(mask)") ||
        !grow(sc.keep, sc.keep_bytes, ((size_t)boxes_num + 1) * 4, "hipMalloc(keep)") ||
        !ok(hipMemcpy(sc.boxes, boxes_host, bytes, hipMemcpyHostToDevice), "hipMemcpy(boxes)"))
        return 3;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(col_blocks, col_blocks), dim3(64), 0, 0, sc.boxes, boxes_num, boxes_dim,
                       nms_overlap_thresh, sc.mask, col_blocks);
    if (col_blocks <= 64)
        hipLaunchKernelGGL(nms_sweep_kernel, dim3(1), dim3(64), 0, 0, sc.mask, boxes_num, col_blocks, sc.keep, sc.keep + boxes_num);
    else
        hipLaunchKernelGGL(nms_sweep_lds_kernel, dim3(1), dim3(64), (size_t)col_blocks * 8, 0, sc.mask, boxes_num, col_blocks, sc.keep,
                           sc.keep + boxes_num);
    // one copy brings the count and the indices (kept <= boxes_num entries are meaningful)
    if (!ok(hipGetLastError(), "nms launch") ||
        !ok(hipMemcpy(num_out, sc.keep + boxes_num, 4, hipMemcpyDeviceToHost), "hipMemcpy(num_out)") ||
        !ok(hipMemcpy(keep_out, sc.keep, (size_t)(*num_out) * 4, hipMemcpyDeviceToHost), "hipMemcpy(keep)"))
        return 4;
    return 0;
}

// frees the per-device scratch (boxes, mask, kept indices) hrn_nms keeps between calls; device_id < 0: every device
extern "C" int hrn_nms_release(int device_id) {
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (int d = 0; d < kMaxDevices; ++d) {
        if (device_id >= 0 && d != device_id) continue;
        Scratch &sc = g_scratch[d];
        std::lock_guard<std::mutex> lock(sc.mu);
        if (!sc.boxes && !sc.mask && !sc.keep) continue;
        if (hipSetDevice(d) != hipSuccess) continue;
        if (sc.boxes) (void)hipFree(sc.boxes);
        if (sc.mask) (void)hipFree(sc.mask);
        if (sc.keep) (void)hipFree(sc.keep);
        sc.boxes = nullptr, sc.mask = nullptr, sc.keep = nullptr;
        sc.boxes_bytes = sc.mask_bytes = sc.keep_bytes = 0;
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    return 0;
}
