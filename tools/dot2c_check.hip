// checks that v_dot2c_f32_bf16 with a (1,0) / (0,1) selector equals acc + bf16->f32(res half) bit for bit,
// and times it against the shift/and + add pair.   hipcc --offload-arch=gfx950 -O3 -o dot2c_check dot2c_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
__global__ void check(const float* acc, const unsigned* res, float* out_dot, float* out_ref, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned r = res[i];
    float a = acc[i], b = acc[i];
    const unsigned sel_lo = 0x00003f80u, sel_hi = 0x3f800000u;
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a) : "s"(sel_lo), "v"(r));
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(b) : "s"(sel_hi), "v"(r));
    out_dot[2 * i] = a, out_dot[2 * i + 1] = b;
    out_ref[2 * i] = acc[i] + __uint_as_float(r << 16);
    out_ref[2 * i + 1] = acc[i] + __uint_as_float(r & 0xffff0000u);
}
int main() {
    const int n = 1 << 20;
    float* ha = (float*)malloc(n * 4); unsigned* hr = (unsigned*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) {
        float f = ((rand() % 20001) - 10000) / 977.0f * (i % 7 == 0 ? 1e-3f : 1.f);
        if (i % 1000 == 0) f = 0.f;
        ha[i] = f;
        float x = ((rand() % 20001) - 10000) / 613.0f, y = ((rand() % 20001) - 10000) / 3011.0f;
        if (i % 500 == 0) x = 0.f;
        if (i % 333 == 0) y = 1e-40f;  // denormal after truncation -> bf16 denormal
        unsigned ux, uy; memcpy(&ux, &x, 4); memcpy(&uy, &y, 4);
        hr[i] = (ux >> 16) | (uy & 0xffff0000u);
    }
    float *da, *dd, *dr; unsigned* dres;
    hipMalloc(&da, n * 4); hipMalloc(&dres, n * 4); hipMalloc(&dd, n * 8); hipMalloc(&dr, n * 8);
    hipMemcpy(da, ha, n * 4, hipMemcpyHostToDevice); hipMemcpy(dres, hr, n * 4, hipMemcpyHostToDevice);
    check<<<n / 256, 256>>>(da, dres, dd, dr, n);
    float* hd = (float*)malloc(n * 8); float* hf = (float*)malloc(n * 8);
    hipMemcpy(hd, dd, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hf, dr, n * 8, hipMemcpyDeviceToHost);
    long bad = 0; double maxrel = 0;
    for (int i = 0; i < 2 * n; ++i)
        if (memcmp(&hd[i], &hf[i], 4)) {
            ++bad;
            double rel = fabs((double)hd[i] - hf[i]) / (fabs((double)hf[i]) + 1e-30);
            if (rel > maxrel) maxrel = rel;
            if (bad <= 5) printf("  diff at %d: dot %.9g ref %.9g (acc %.9g res %08x)\n", i, hd[i], hf[i], ha[i / 2], hr[i / 2]);
        }
    printf("dot2c vs add: %ld of %d differ, max rel %.3g\n", bad, 2 * n, maxrel);
    return 0;
}
