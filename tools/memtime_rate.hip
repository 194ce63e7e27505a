// calibrates s_memtime: ticks per second, idle and under MFMA load
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void spin(long long ticks, long long* out) {
    long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) {}
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}
__global__ __launch_bounds__(256) void mfma_timed(long long* out, int iters) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(threadIdx.x + i); y[i] = (__bf16)(float)(i * 3 + 1); }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n v_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\n v_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\n v_mfma_f32_16x16x32_bf16 %3, %4, %5, %3" : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3) : "v"(x), "v"(y));
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (long long)(a0[0] + a1[0] + a2[0] + a3[0]); }
}
int main() {
    long long* d; hipMalloc(&d, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    spin<<<1, 64>>>(1000000, d); hipDeviceSynchronize();
    hipEventRecord(e0); spin<<<1, 64>>>(100000000LL, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("idle spin: 1e8 ticks in %.3f ms -> %.1f MHz\n", ms, 1e8 / ms / 1e3);
    const int iters = 200000;
    mfma_timed<<<256, 256>>>(d, 1000); hipDeviceSynchronize();
    hipEventRecord(e0); mfma_timed<<<256, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("mfma load: %lld ticks in %.3f ms -> %.1f MHz tick rate; %.2f ticks per MFMA per wave; %.2f ns per MFMA\n", h[0], ms, h[0] / ms / 1e3,
           (double)h[0] / (iters * 4.0), ms * 1e6 / (iters * 4.0));
    return 0;
}
