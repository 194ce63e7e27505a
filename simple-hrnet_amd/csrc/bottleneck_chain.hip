// Bottleneck tail + next Bottleneck head in one pass (layer1, modules.py:20-40), bf16, gfx950.
//
//   y  = relu( W3 * t2 + b3 + residual )        conv3 + bn3 + shortcut + ReLU of block b      (64 -> 256)
//   t' = relu( W1' * y + b1' )                  conv1 + bn1 + ReLU of block b+1              (256 -> 64)
//
// Both are 1x1 convolutions whose cost is memory: run separately, y (0.9 GB per 256 crops) is written by the first
// and read back by the second.  Here a wave keeps its 16 pixels x 256 channels of y in registers: the D fragments
// of the first product, rounded to bf16 exactly as they are stored, ARE the B operand of the second one.
//   * W3 is packed with 32-cout groups (generic image, NR = 2): lane (li, g) then owns channels j*32 + g*8 + [0,8)
//     of pixel li for j = 0..7 -- which is the K slice (chunk j, k-group g) the second MFMA wants from that lane.
//     No cross-lane movement, no LDS round trip, natural K order (bit-identical to the two separate kernels).
//   * W1' uses the generic image with NR = 4: a lane ends with 16 contiguous output channels.
//   * Both weight images (32 KiB each) and the biases sit in LDS; every MFMA reads its A fragment from there.
//   * Block 0's shortcut is itself a 1x1 conv of the block input (downsample, 64 -> 256, no ReLU).  The DS variant
//     computes it here, in its own accumulators, rounds it to bf16 exactly where the separate kernel would store
//     it, and adds it -- the 0.9 GB tensor is neither written nor read.
//   * No barrier after the weights are staged: each wave walks its own 16-pixel fragments (2 blocks = 8 waves per
//     CU keep ~80 KiB of loads in flight, which is what the HBM pipe needs).
#include <stdlib.h>

#include "kernels.h"

namespace hrn {

#define GLOBAL_AS __attribute__((address_space(1)))

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

constexpr int CIN = 64, CMID = 256, COUT = 64;
constexpr int W3_BYTES = CMID * CIN * 2, W1_BYTES = COUT * CMID * 2;
constexpr int WDS_BYTES = CMID * CIN * 2;
constexpr int lds_bytes(bool ds) { return W3_BYTES + W1_BYTES + CMID * 4 + COUT * 4 + (ds ? WDS_BYTES + CMID * 4 : 0); }

__device__ __forceinline__ float bf16_f32(short h) { return __uint_as_float(((unsigned)(unsigned short)h) << 16); }
__device__ __forceinline__ short f32_bf16(float f) {  // round to nearest even, as every other store of the engine
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (short)(u >> 16);
}

}  // namespace

// DS = false: 4 waves, 66 KiB of LDS, two blocks per CU.  DS = true: 8 waves share 99 KiB, one block per CU.
template <bool DS>
__global__ __launch_bounds__(DS ? 512 : 256, DS ? 1 : 2) void bottleneck_chain_kernel(const ChainArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = DS ? 512 : 256, WAVES = NT / 64;
    constexpr int DS_OFF = W3_BYTES + W1_BYTES + CMID * 4 + COUT * 4;  // [wds image][bds]
    {
        const uint4 *w3 = (const uint4 *)p.w3, *w1 = (const uint4 *)p.w1;
        uint4 *d3 = (uint4 *)smem, *d1 = (uint4 *)(smem + W3_BYTES);
        for (int i = threadIdx.x; i < W3_BYTES / 16; i += NT) d3[i] = w3[i];
        for (int i = threadIdx.x; i < W1_BYTES / 16; i += NT) d1[i] = w1[i];
        float *b3 = (float *)(smem + W3_BYTES + W1_BYTES);
        for (int i = threadIdx.x; i < CMID; i += NT) b3[i] = p.b3[i];
        if (threadIdx.x < COUT) b3[CMID + threadIdx.x] = p.b1[threadIdx.x];
        if constexpr (DS) {
            const uint4 *wd = (const uint4 *)p.wds;
            uint4 *dd = (uint4 *)(smem + DS_OFF);
            for (int i = threadIdx.x; i < WDS_BYTES / 16; i += NT) dd[i] = wd[i];
            float *bd = (float *)(smem + DS_OFF + WDS_BYTES);
            for (int i = threadIdx.x; i < CMID; i += NT) bd[i] = p.bds[i];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const s16x8 *w3f = (const s16x8 *)smem + lane;                     // fragment (f, c) at [(f*2 + c)*64]
    const s16x8 *w1f = (const s16x8 *)(smem + W3_BYTES) + lane;        // fragment (f, kc) at [(f*8 + kc)*64]
    const float *b3s = (const float *)(smem + W3_BYTES + W1_BYTES);

    const GLOBAL_AS short *__restrict__ in = (const GLOBAL_AS short *)p.in;
    const GLOBAL_AS short *__restrict__ res = (const GLOBAL_AS short *)p.res;
    GLOBAL_AS short *__restrict__ out_y = (GLOBAL_AS short *)p.out_y;
    GLOBAL_AS short *__restrict__ out_t = (GLOBAL_AS short *)p.out_t;

    const s16x8 *wdf = (const s16x8 *)(smem + DS_OFF) + lane;          // downsample fragment (f, c), as W3
    const float *bds = (const float *)(smem + DS_OFF + WDS_BYTES);
    const GLOBAL_AS short *__restrict__ xin = (const GLOBAL_AS short *)p.x;

    const int mfrags = (p.m + 15) >> 4;
    for (int mf0 = blockIdx.x * WAVES + wave; mf0 < mfrags; mf0 += gridDim.x * WAVES) {
        const int mf = p.rev ? mfrags - 1 - mf0 : mf0;
        const int q = mf * 16 + li;
        const bool live = q < p.m;
        const int qc = live ? q : 0;
        const int rem = qc % p.hpwp;
        const int ho = rem / p.wp, wo = rem - ho * p.wp;
        const bool ok = live && ho < p.h && wo < p.w;

        s16x8 a[2], r[8];
#pragma unroll
        for (int c = 0; c < 2; ++c) a[c] = *(const GLOBAL_AS s16x8 *)(in + (size_t)qc * CIN + c * 32 + g * 8);
        s16x8 wb[2][4];
        if constexpr (DS) {
            // shortcut = Wds * x + bds, rounded to bf16 (zero on pad pixels) like the tensor it replaces
            s16x8 xa[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) xa[c] = *(const GLOBAL_AS s16x8 *)(xin + (size_t)qc * CIN + c * 32 + g * 8);
            f32x4 accd[16];
#pragma unroll
            for (int f = 0; f < 16; ++f) accd[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) wb[0][t] = wdf[(t * 2 + 0) * 64];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b + 1 < 8) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) wb[(b + 1) & 1][t] = wdf[((4 * ((b + 1) & 3) + t) * 2 + ((b + 1) >> 2)) * 64];
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int f = 4 * (b & 3) + t;
                    accd[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wb[b & 1][t]),
                                                                      __builtin_bit_cast(bf16x8, xa[b >> 2]), accd[f], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 ba = *(const f32x4 *)(bds + j * 32 + g * 8), bb = *(const f32x4 *)(bds + j * 32 + g * 8 + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = accd[2 * j + (e >> 2)][e & 3] + (e < 4 ? ba[e & 3] : bb[e & 3]);
                    if (!ok) v = 0.f;
                    r[j][e] = f32_bf16(v);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = *(const GLOBAL_AS s16x8 *)(res + (size_t)qc * CMID + j * 32 + g * 8);
        }

        // ---- y = W3 * t2: 16 cout fragments x 2 K chunks.  A fragments come from LDS four at a time, one batch
        //      ahead of the MFMAs that use them; sched_barrier keeps hipcc from hoisting all 64 reads (256 VGPRs)
        f32x4 acc[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) wb[0][t] = w3f[(t * 2 + 0) * 64];
#pragma unroll
        for (int b = 0; b < 8; ++b) {  // batch b: chunk c = b >> 2, fragments 4*(b & 3) .. +3
            if (b + 1 < 8) {
#pragma unroll
                for (int t = 0; t < 4; ++t) wb[(b + 1) & 1][t] = w3f[((4 * ((b + 1) & 3) + t) * 2 + ((b + 1) >> 2)) * 64];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int f = 4 * (b & 3) + t;
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wb[b & 1][t]),
                                                                 __builtin_bit_cast(bf16x8, a[b >> 2]), acc[f], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue of conv3: bias, shortcut, ReLU, pad mask; the bf16 image is stored AND kept as operand
        s16x8 y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 ba = *(const f32x4 *)(b3s + j * 32 + g * 8), bb = *(const f32x4 *)(b3s + j * 32 + g * 8 + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = acc[2 * j + (e >> 2)][e & 3] + (e < 4 ? ba[e & 3] : bb[e & 3]);
                v += bf16_f32(r[j][e]);
                v = fmaxf(v, 0.f);
                if (!ok) v = 0.f;
                y[j][e] = f32_bf16(v);
            }
            if (live) *(GLOBAL_AS s16x8 *)(out_y + (size_t)q * CMID + j * 32 + g * 8) = y[j];
        }
        // ---- t' = W1' * y: 4 cout fragments x 8 K chunks, chunk kc's operand is y[kc]
        f32x4 acc2[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) acc2[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) wb[0][t] = w1f[(t * 8 + 0) * 64];
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            if (kc + 1 < 8) {
#pragma unroll
                for (int t = 0; t < 4; ++t) wb[(kc + 1) & 1][t] = w1f[(t * 8 + kc + 1) * 64];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wb[kc & 1][t]),
                                                                  __builtin_bit_cast(bf16x8, y[kc]), acc2[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            s16x8 o8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = acc2[2 * h + (e >> 2)][e & 3] + b3s[CMID + g * 16 + h * 8 + e];
                v = fmaxf(v, 0.f);
                if (!ok) v = 0.f;
                o8[e] = f32_bf16(v);
            }
            if (live) *(GLOBAL_AS s16x8 *)(out_t + (size_t)q * COUT + g * 16 + h * 8) = o8;
        }
    }
}

template <bool DS>
static hipError_t launch_chain_t(const ChainArgs &a, hipStream_t s) {
    static std::atomic<unsigned long long> lds_set{0};   // per device: kernels.h set_dynamic_lds
    {
        const hipError_t e = set_dynamic_lds((const void *)bottleneck_chain_kernel<DS>, lds_bytes(DS), lds_set);
        if (e != hipSuccess) return e;
    }
    // persistent: 8 waves per CU re-use their staged weights over many 16-pixel fragments
    const int blocks_env = a.max_blocks > 0 ? a.max_blocks : 512;
    constexpr int WAVES = DS ? 8 : 4;
    const int mfrags = (a.m + 15) / 16;
    int blocks = (mfrags + WAVES - 1) / WAVES;
    const int cap = DS ? blocks_env / 2 : blocks_env;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(bottleneck_chain_kernel<DS>, dim3(blocks), dim3(64 * WAVES), lds_bytes(DS), s, a);
    return hipGetLastError();
}

hipError_t launch_bottleneck_chain(const ChainArgs &a, hipStream_t s) {
    if (a.m <= 0) return hipSuccess;
    return a.wds ? launch_chain_t<true>(a, s) : launch_chain_t<false>(a, s);
}

}  // namespace hrn
