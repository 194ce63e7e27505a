// Crop pre-path of SimpleHRNet.predict(), single image / multi-person (SimpleHRNet.py:236-278), on the GPU:
// the frame crosses PCIe once as uint8 and every person's crop is cut, zero-padded, resized and normalised here,
// straight into the (n,3,H,W) fp32 batch the stem reads.
//
// The resize is Pillow's (torchvision Resize on a PIL image = Image.resize(BILINEAR); libImaging/Resample.c):
//   two separable 8-bit passes, horizontal then vertical, rounded and clipped to uint8 in between;
//   per output sample: center = (xx + 0.5) * scale, support = max(scale, 1), taps [xmin, xmin + n),
//   weights = triangle((x + xmin - center + 0.5) / max(scale, 1)) normalised to sum 1 in double, then
//   22-bit fixed point: kk = (int)(0.5 + k * 2^22); sample = clip8((2^21 + sum src * kk) >> 22).
// The coefficients are recomputed per thread in double with the same operation order (fp contraction off), so the
// result is bit-identical to Pillow's; ToTensor / Normalize are the float32 operations torchvision performs
// (v / 255, then (x - mean) / std).  A pass whose size does not change is the identity, as in Pillow.
#include "kernels.h"

namespace hrn {

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

struct Taps {
    int xmin, n;
    double scale, support, ss;
};

#pragma clang fp contract(off)
__device__ __forceinline__ Taps taps_of(int xx, int in_size, int out_size) {
    Taps t;
    t.scale = (double)in_size / out_size;
    const double filterscale = t.scale < 1.0 ? 1.0 : t.scale;
    t.support = 1.0 * filterscale;
    t.ss = 1.0 / filterscale;
    const double center = (xx + 0.5) * t.scale;
    int xmin = (int)(center - t.support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + t.support + 0.5);
    if (xmax > in_size) xmax = in_size;
    t.xmin = xmin, t.n = xmax - xmin;
    return t;
}

#pragma clang fp contract(off)
__device__ __forceinline__ double tri(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

// weight k of output sample xx, normalised and quantised exactly as precompute_coeffs + normalize_coeffs_8bpc
#pragma clang fp contract(off)
__device__ __forceinline__ void weights_of(const Taps &t, int xx, double *ww_out) {
    const double center = (xx + 0.5) * t.scale;
    double ww = 0.0;
    for (int x = 0; x < t.n; ++x) ww += tri((x + t.xmin - center + 0.5) * t.ss);
    *ww_out = ww;
}
#pragma clang fp contract(off)
__device__ __forceinline__ int coeff(const Taps &t, int xx, int x, double ww) {
    const double center = (xx + 0.5) * t.scale;
    double k = tri((x + t.xmin - center + 0.5) * t.ss);
    if (ww != 0.0) k /= ww;
    return (int)(0.5 + k * (double)(1 << PRECISION_BITS));
}

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// padded-crop pixel (y, x) of channel c (RGB order) -- zero in the padding, the BGR frame flipped elsewhere
__device__ __forceinline__ int crop_px(const unsigned char *frame, int frame_w, const CropParams &cp, int y, int x, int c) {
    const int yy = y - cp.pad_top, xx = x - cp.pad_left;
    if (yy < 0 || yy >= cp.h_crop || xx < 0 || xx >= cp.w_crop) return 0;
    return frame[((size_t)(cp.y1 + yy) * frame_w + cp.x1 + xx) * 3 + (2 - c)];
}

}  // namespace

// pass 1: rows of the padded crop -> W output columns (uint8 RGB, row-major [h_pad][W][3] per crop)
__global__ __launch_bounds__(256) void prepath_horizontal_kernel(const unsigned char *frame, int frame_w,
                                                                 const CropParams *crops, unsigned char *tmp, int W) {
    const CropParams cp = crops[blockIdx.y];
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)cp.h_pad * W) return;
    const int y = (int)(idx / W), xx = (int)(idx - (long)y * W);
    unsigned char *o = tmp + cp.tmp_off + ((size_t)y * W + xx) * 3;
    if (cp.w_pad == W) {  // no horizontal pass in Pillow either
        for (int c = 0; c < 3; ++c) o[c] = (unsigned char)crop_px(frame, frame_w, cp, y, xx, c);
        return;
    }
    const Taps t = taps_of(xx, cp.w_pad, W);
    double ww;
    weights_of(t, xx, &ww);
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < t.n; ++x) {
        const int k = coeff(t, xx, x, ww);
        s0 += crop_px(frame, frame_w, cp, y, t.xmin + x, 0) * k;
        s1 += crop_px(frame, frame_w, cp, y, t.xmin + x, 1) * k;
        s2 += crop_px(frame, frame_w, cp, y, t.xmin + x, 2) * k;
    }
    o[0] = (unsigned char)clip8(s0), o[1] = (unsigned char)clip8(s1), o[2] = (unsigned char)clip8(s2);
}

// pass 2: h_pad rows -> H output rows, then ToTensor + Normalize into the (n,3,H,W) fp32 batch
__global__ __launch_bounds__(256) void prepath_vertical_kernel(const CropParams *crops, const unsigned char *tmp, float *images,
                                                               int H, int W) {
    const CropParams cp = crops[blockIdx.y];
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)H * W) return;
    const int yy = (int)(idx / W), xx = (int)(idx - (long)yy * W);
    const unsigned char *src = tmp + cp.tmp_off;
    int v[3];
    if (cp.h_pad == H) {
        for (int c = 0; c < 3; ++c) v[c] = src[((size_t)yy * W + xx) * 3 + c];
    } else {
        const Taps t = taps_of(yy, cp.h_pad, H);
        double ww;
        weights_of(t, yy, &ww);
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int y = 0; y < t.n; ++y) {
            const int k = coeff(t, yy, y, ww);
            const unsigned char *px = src + ((size_t)(t.xmin + y) * W + xx) * 3;
            s0 += px[0] * k, s1 += px[1] * k, s2 += px[2] * k;
        }
        v[0] = clip8(s0), v[1] = clip8(s1), v[2] = clip8(s2);
    }
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};  // SimpleHRNet.py:171
    float *o = images + (size_t)blockIdx.y * 3 * H * W + (size_t)yy * W + xx;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float x = (float)v[c] / 255.0f;           // ToTensor
        o[(size_t)c * H * W] = (x - mean[c]) / stdv[c];  // Normalize
    }
}

hipError_t launch_prepath(const unsigned char *frame_dev, int frame_w, const CropParams *crops_dev, int n, int max_h_pad,
                          unsigned char *tmp_dev, float *images_dev, int H, int W, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    dim3 g1((unsigned)(((long)max_h_pad * W + 255) / 256), n);
    hipLaunchKernelGGL(prepath_horizontal_kernel, g1, dim3(256), 0, s, frame_dev, frame_w, crops_dev, tmp_dev, W);
    dim3 g2((unsigned)(((long)H * W + 255) / 256), n);
    hipLaunchKernelGGL(prepath_vertical_kernel, g2, dim3(256), 0, s, crops_dev, tmp_dev, images_dev, H, W);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Single-person pre-path: cv2.resize(frame, (W, H), interpolation) + BGR -> RGB + ToTensor + Normalize
// (SimpleHRNet.py:213-222, 355-366) for 8-bit 3-channel frames.  OpenCV is a third-party dependency of the reference that is
// not in this image: the arithmetic below follows the published generic path of modules/imgproc/src/resize.cpp (fixed-point
// coefficients of 11 bits, int32 passes; see oracle/cv2_resize_oracle.py, which this kernel matches bit for bit) -- parity
// with cv2 itself is UNPINNED.  It is OpenCV's SCALAR path: the SIMD builds of the stock wheels run the cubic vertical pass in
// float32 (VResizeCubicVec_32s8u) and can differ from it by one grey level on some pixels;
// tests/golden/make_cv2_golden.py makes the pin wherever opencv-python is installed.
// Tap tables are formed on the device (no host staging): one thread per output column / row.
__global__ __launch_bounds__(256) void resize_taps_kernel(int src_w, int src_h, int W, int H, double scale_x, double scale_y,
                                                          int interp, ResizeTaps *taps) {
#pragma clang fp contract(off)
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= W + H) return;
    const bool is_x = t < W;
    const int d = is_x ? t : t - W, src = is_x ? src_w : src_h;
    const double scale = is_x ? scale_x : scale_y;
    ResizeTaps o;
    o.ofs = 0, o.c[0] = o.c[1] = o.c[2] = o.c[3] = 0;
    if (interp == 0) {   // resizeNN
        int s = (int)floor(__dmul_rn((double)d, scale));
        o.ofs = s < src - 1 ? s : src - 1, o.c[0] = 2048;
    } else {
        // (no contraction into fused multiply-adds anywhere: the reference rounds after every operation)
        float f = (float)__dsub_rn(__dmul_rn(__dadd_rn((double)d, 0.5), scale), 0.5);
        int s = (int)floorf(f);
        f = __fsub_rn(f, (float)s);
        if (interp == 2) {   // interpolateCubic, A = -0.75
            const float A = -0.75f;
            const float x1 = __fadd_rn(f, 1.f), xm = __fsub_rn(1.f, f);
            const float c0 = __fsub_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fsub_rn(__fmul_rn(A, x1), __fmul_rn(5.f, A)), x1), __fmul_rn(8.f, A)), x1),
                                       __fmul_rn(4.f, A));
            const float c1 = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(A, 2.f), f), __fadd_rn(A, 3.f)), f), f), 1.f);
            const float c2 = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(A, 2.f), xm), __fadd_rn(A, 3.f)), xm), xm), 1.f);
            const float c3 = __fsub_rn(__fsub_rn(__fsub_rn(1.f, c0), c1), c2);
            const float c[4] = {c0, c1, c2, c3};
            o.ofs = s - 1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int v = __float2int_rn(__fmul_rn(c[k], 2048.f));   // saturate_cast<short>: nearest, ties to even
                o.c[k] = (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v);
            }
        } else {             // INTER_LINEAR
            if (is_x && s < 0) s = 0, f = 0.f;
            if (is_x && s >= src - 1) s = src - 1, f = 0.f;
            o.ofs = s;
            o.c[0] = (short)__float2int_rn(__fmul_rn(__fsub_rn(1.f, f), 2048.f));
            o.c[1] = (short)__float2int_rn(__fmul_rn(f, 2048.f));
        }
    }
    taps[t] = o;
}

__global__ __launch_bounds__(256) void resize_frames_kernel(const unsigned char *frames, int src_h, int src_w, const ResizeTaps *taps,
                                                            int interp, float *images, int H, int W) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)H * W) return;
    const int yy = (int)(idx / W), xx = (int)(idx - (long)yy * W);
    const unsigned char *src = frames + (size_t)blockIdx.y * src_h * src_w * 3;
    const ResizeTaps tx = taps[xx], ty = taps[W + yy];
    const int K = interp == 2 ? 4 : interp == 1 ? 2 : 1;
    int hor[4][3];
    for (int r = 0; r < K; ++r) {
        int y = ty.ofs + r;
        y = y < 0 ? 0 : y > src_h - 1 ? src_h - 1 : y;                 // replicate border
        int s0 = 0, s1 = 0, s2 = 0;
        for (int k = 0; k < K; ++k) {
            int x = tx.ofs + k;
            x = x < 0 ? 0 : x > src_w - 1 ? src_w - 1 : x;
            const unsigned char *px = src + ((size_t)y * src_w + x) * 3;
            const int a = tx.c[k];
            s0 += px[0] * a, s1 += px[1] * a, s2 += px[2] * a;
        }
        hor[r][0] = s0, hor[r][1] = s1, hor[r][2] = s2;
    }
    int v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int o;
        if (interp == 2) {        // VResizeCubic + FixedPtCast<int, uchar, 22>
            o = (hor[0][c] * ty.c[0] + hor[1][c] * ty.c[1] + hor[2][c] * ty.c[2] + hor[3][c] * ty.c[3] + (1 << 21)) >> 22;
        } else if (interp == 1) { // VResizeLinear<uchar, int, short>
            o = (((ty.c[0] * (hor[0][c] >> 4)) >> 16) + ((ty.c[1] * (hor[1][c] >> 4)) >> 16) + 2) >> 2;
        } else {
            o = hor[0][c] >> 11;
        }
        v[c] = o < 0 ? 0 : o > 255 ? 255 : o;
    }
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};  // SimpleHRNet.py:171
    float *o = images + (size_t)blockIdx.y * 3 * H * W + (size_t)yy * W + xx;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float x = (float)v[2 - c] / 255.0f;       // BGR -> RGB, ToTensor
        o[(size_t)c * H * W] = (x - mean[c]) / stdv[c];  // Normalize
    }
}

hipError_t launch_resize_frames(const unsigned char *frames_dev, int n, int src_h, int src_w, int interp, ResizeTaps *taps_dev,
                                float *images_dev, int H, int W, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const double scale_x = 1.0 / ((double)W / (double)src_w), scale_y = 1.0 / ((double)H / (double)src_h);
    hipLaunchKernelGGL(resize_taps_kernel, dim3((W + H + 255) / 256), dim3(256), 0, s, src_w, src_h, W, H, scale_x, scale_y, interp, taps_dev);
    dim3 g((unsigned)(((long)H * W + 255) / 256), n);
    hipLaunchKernelGGL(resize_frames_kernel, g, dim3(256), 0, s, frames_dev, src_h, src_w, taps_dev, interp, images_dev, H, W);
    return hipGetLastError();
}

}  // namespace hrn
