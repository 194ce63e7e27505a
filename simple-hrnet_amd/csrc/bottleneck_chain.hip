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
//   * Round 5 (C3): the Bottleneck's 3x3 convolution (conv2 + bn2 + ReLU, 64 -> 64) in FRONT of conv3, for the blocks without a
//     projection shortcut:  t2 = relu( W2 (*) t1 + b2 )  is computed per 16-pixel fragment from conv1's output t1 -- the nine taps
//     are nine constant row shifts of the flat padded layout, 18 fragment loads of 16 B per lane that L1 / L2 serve (a pixel's
//     128-byte row is one cache line, neighbouring taps share 15 of their 16 lines) -- in the GENERIC kernel's arithmetic
//     (k = tap * 64 + ci in 32-wide chunks, accumulators from zero, bias added last; W2 in the generic image with NR = 2), so
//     its D fragments, rounded to bf16 exactly where conv_direct_kernel would store them, ARE conv3's B operand: t2 (0.23 GB
//     per 256 crops) is neither written nor read, and the launch of the 3x3 kernel (0.15 ms) is gone.  HBM traffic of the
//     launch is unchanged (t1 is read instead of t2); W2 (72 KiB) joins W3 / W1' in LDS: one 8-wave block per CU.
//     C3 = 2: the last Bottleneck of the layer -- no conv1 of a next block behind it.
#include <stdlib.h>

#include "kernels.h"

namespace hrn {

#define GLOBAL_AS __attribute__((address_space(1)))

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

// C3 variants: 8 waves per block (12 -- the registers would allow three per SIMD -- measured 6 % slower, and so did requesting the
// shortcut before the 3x3 instead of behind it: profiles/EXPERIMENTS.md, round 5)
constexpr int C3_WAVES = 8;
constexpr int CIN = 64, CMID = 256, COUT = 64;
constexpr int W3_BYTES = CMID * CIN * 2, W1_BYTES = COUT * CMID * 2;
constexpr int WDS_BYTES = CMID * CIN * 2;
constexpr int W2_CHUNKS = 18, W2_BYTES = 4 * W2_CHUNKS * 1024;   // conv2: 64 couts = 4 fragments, K = 9 taps x 64 = 18 chunks
constexpr int lds_bytes(bool ds, int c3 = 0) {
    return W3_BYTES + W1_BYTES + CMID * 4 + COUT * 4 + (ds ? WDS_BYTES + CMID * 4 : 0) + (c3 ? W2_BYTES + CIN * 4 : 0);
}

__device__ __forceinline__ float bf16_f32(short h) { return __uint_as_float(((unsigned)(unsigned short)h) << 16); }
__device__ __forceinline__ short f32_bf16(float f) {  // round to nearest even, as every other store of the engine
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (short)(u >> 16);
}

}  // namespace

// DS = false: 4 waves, 66 KiB of LDS, two blocks per CU.  DS = true: 8 waves share 99 KiB, one block per CU.
// C3 = 1 / 2 (DS = false): 8 waves share 138 KiB, one block per CU.
template <bool DS, int C3 = 0>
__global__ __launch_bounds__(C3 ? 64 * C3_WAVES : DS ? 512 : 256, C3 ? C3_WAVES / 4 : DS ? 1 : 2) void bottleneck_chain_kernel(const ChainArgs p) {
    static_assert(!(DS && C3), "the 3x3 front exists for the blocks without a projection shortcut");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = C3 ? 64 * C3_WAVES : DS ? 512 : 256, WAVES = NT / 64;
    constexpr int DS_OFF = W3_BYTES + W1_BYTES + CMID * 4 + COUT * 4;  // [wds image][bds] / C3: [w2 image][b2]
    {
        const uint4 *w3 = (const uint4 *)p.w3, *w1 = (const uint4 *)p.w1;
        uint4 *d3 = (uint4 *)smem, *d1 = (uint4 *)(smem + W3_BYTES);
        for (int i = threadIdx.x; i < W3_BYTES / 16; i += NT) d3[i] = w3[i];
        if constexpr (C3 != 2)
            for (int i = threadIdx.x; i < W1_BYTES / 16; i += NT) d1[i] = w1[i];
        float *b3 = (float *)(smem + W3_BYTES + W1_BYTES);
        for (int i = threadIdx.x; i < CMID; i += NT) b3[i] = p.b3[i];
        if constexpr (C3 != 2)
            if (threadIdx.x < COUT) b3[CMID + threadIdx.x] = p.b1[threadIdx.x];
        if constexpr (DS) {
            const uint4 *wd = (const uint4 *)p.wds;
            uint4 *dd = (uint4 *)(smem + DS_OFF);
            for (int i = threadIdx.x; i < WDS_BYTES / 16; i += NT) dd[i] = wd[i];
            float *bd = (float *)(smem + DS_OFF + WDS_BYTES);
            for (int i = threadIdx.x; i < CMID; i += NT) bd[i] = p.bds[i];
        }
        if constexpr (C3 != 0) {
            const uint4 *w2 = (const uint4 *)p.w2;
            uint4 *d2 = (uint4 *)(smem + DS_OFF);
            for (int i = threadIdx.x; i < W2_BYTES / 16; i += NT) d2[i] = w2[i];
            float *b2 = (float *)(smem + DS_OFF + W2_BYTES);
            if (threadIdx.x < CIN) b2[threadIdx.x] = p.b2[threadIdx.x];
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
        s16x8 wb[2][4];
        if constexpr (C3 == 0) {
#pragma unroll
            for (int c = 0; c < 2; ++c) a[c] = *(const GLOBAL_AS s16x8 *)(in + (size_t)qc * CIN + c * 32 + g * 8);
        } else {
            // ---- t2 = relu(W2 (*) t1 + b2): 4 cout fragments x 18 K chunks (chunk kc = tap kc / 2, channels (kc & 1) * 32 + 8 g ..).
            //      The tap fragments of one kernel row (6 chunks) are requested while the previous row's MFMAs run; the guard
            //      rows of the buffer make every shifted row a valid address (pad positions hold zeros: the 3x3's zero padding).
            const s16x8 *w2f = (const s16x8 *)(smem + DS_OFF) + lane;   // fragment (f, kc) at [(f * 18 + kc) * 64]
            const float *b2s = (const float *)(smem + DS_OFF + W2_BYTES);
            const GLOBAL_AS short *__restrict__ t1 = (const GLOBAL_AS short *)p.in3;
            const GLOBAL_AS short *t1q = t1 + ((long)qc - p.wp - 1) * CIN + g * 8;   // tap (0, 0) of this lane's pixel
            s16x8 xt[2][6];
#pragma unroll
            for (int e = 0; e < 6; ++e) xt[0][e] = *(const GLOBAL_AS s16x8 *)(t1q + (long)(e >> 1) * CIN + (e & 1) * 32);
            f32x4 acc2[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) acc2[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) {
                if (dh + 1 < 3) {
#pragma unroll
                    for (int e = 0; e < 6; ++e)
                        xt[(dh + 1) & 1][e] = *(const GLOBAL_AS s16x8 *)(t1q + ((long)(dh + 1) * p.wp + (e >> 1)) * CIN + (e & 1) * 32);
                }
#pragma unroll
                for (int e = 0; e < 6; ++e) {
                    const int kc = dh * 6 + e;
#pragma unroll
                    for (int f = 0; f < 4; ++f)
                        acc2[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2f[(f * W2_CHUNKS + kc) * 64]),
                                                                          __builtin_bit_cast(bf16x8, xt[dh & 1][e]), acc2[f], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // bias, ReLU, zero on pad pixels, round to bf16 -- conv_direct_kernel's epilogue, value for value; fragment f = 2 c + h
            // row 4 g + r holds channel 32 c + 8 g + 4 h + r (generic image, NR = 2): the lane's eight values of K chunk c of conv3
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = acc2[2 * c + (e >> 2)][e & 3] + b2s[c * 32 + g * 8 + e];
                    v = fmaxf(v, 0.f);
                    if (!ok) v = 0.f;
                    a[c][e] = f32_bf16(v);
                }
        }
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
        if constexpr (C3 == 2) continue;   // the layer's last Bottleneck: nothing behind conv3
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

template <bool DS, int C3 = 0>
static hipError_t launch_chain_t(const ChainArgs &a, hipStream_t s) {
    static std::atomic<unsigned long long> lds_set{0};   // per device: kernels.h set_dynamic_lds
    {
        const hipError_t e = set_dynamic_lds((const void *)bottleneck_chain_kernel<DS, C3>, lds_bytes(DS, C3), lds_set);
        if (e != hipSuccess) return e;
    }
    // persistent: 8 waves per CU re-use their staged weights over many 16-pixel fragments
    const int blocks_env = a.max_blocks > 0 ? a.max_blocks : 512;
    constexpr int WAVES = C3 ? C3_WAVES : DS ? 8 : 4;
    const int mfrags = (a.m + 15) / 16;
    int blocks = (mfrags + WAVES - 1) / WAVES;
    const int cap = (DS || C3) ? (blocks_env / 2 > 1 ? blocks_env / 2 : 1) : blocks_env;   // (HRN_CHAIN_BLOCKS < 2 must not give a zero grid)
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((bottleneck_chain_kernel<DS, C3>), dim3(blocks), dim3(64 * WAVES), lds_bytes(DS, C3), s, a);
    return hipGetLastError();
}

hipError_t launch_bottleneck_chain(const ChainArgs &a, hipStream_t s) {
    if (a.m <= 0) return hipSuccess;
    if (a.w2) return a.w1 ? launch_chain_t<false, 1>(a, s) : launch_chain_t<false, 2>(a, s);
    return a.wds ? launch_chain_t<true>(a, s) : launch_chain_t<false>(a, s);
}

}  // namespace hrn
