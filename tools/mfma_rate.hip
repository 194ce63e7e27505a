// Micro-benchmark: issue rate of the gfx950 MFMA shapes used by the conv kernels (one wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    f32x16 b0 = {}, b1 = {};
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(threadIdx.x + i); y[i] = (__bf16)(float)(i * 3 + 1); }
    s16x4 xs = {1, 2, 3, 4}, ys = {5, 6, 7, 8};
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n v_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\n v_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\n v_mfma_f32_16x16x32_bf16 %3, %4, %5, %3" : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3) : "v"(x), "v"(y));
        } else if constexpr (MODE == 1) {
            asm volatile("v_mfma_f32_16x16x16_bf16 %0, %4, %5, %0\n v_mfma_f32_16x16x16_bf16 %1, %4, %5, %1\n v_mfma_f32_16x16x16_bf16 %2, %4, %5, %2\n v_mfma_f32_16x16x16_bf16 %3, %4, %5, %3" : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3) : "v"(xs), "v"(ys));
        } else if constexpr (MODE == 2) {
            b0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, b0, 0, 0, 0);
            b1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, b1, 0, 0, 0);
            b0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, b0, 0, 0, 0);
            b1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, b1, 0, 0, 0);
        } else {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, 2.f, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, 2.f, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, 2.f, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, 2.f, a3, 0, 0, 0);
        }
    }
    float s = a0[0] + a1[1] + a2[2] + a3[3] + b0[0] + b1[5];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, double flop_per_mfma, float* d) {
    const int iters = 100000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 100);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n = (double)blocks * 4 * iters * 4;  // waves * iters * mfma per iter
    double cyc = ms * 1e-3 * 2.4e9 / (iters * 4.0);
    printf("%-14s %8.3f ms  %8.1f TFLOP/s   ~%.1f cycles/MFMA/SIMD (at 2.4GHz)\n", name, ms, n * flop_per_mfma / ms / 1e9, cyc);
}
int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    for (int rep = 0; rep < 3; ++rep) run<2>("32x32x16 bf16", 2.0 * 32 * 32 * 16, d);
    run<0>("16x16x32 bf16", 2.0 * 16 * 16 * 32, d);
    run<0>("16x16x32 bf16", 2.0 * 16 * 16 * 32, d);
    run<1>("16x16x16 bf16", 2.0 * 16 * 16 * 16, d);
    run<2>("32x32x16 bf16", 2.0 * 32 * 32 * 16, d);
    run<3>("16x16x4 f32", 2.0 * 16 * 16 * 4, d);
    return 0;
}
