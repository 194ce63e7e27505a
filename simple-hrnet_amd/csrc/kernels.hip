// HIP kernels of the HRNet hot path for gfx950 (MI355X, CDNA4).  Hand-written; wave = 64.
//
//   conv_direct_kernel  generic implicit-GEMM convolution (1x1 / 3x3, stride 1 / 2) on MFMA with the
//                       fused epilogue  out = [relu]( acc + bias [+ residual] ), zero at pad pixels.
//                       bf16: v_mfma_f32_16x16x32_bf16 (fp32 accumulate); fp32: v_mfma_f32_16x16x4_f32
//                       (exact fp32 fma chain).  Operands are swapped (D = W * X^T) so that one lane owns
//                       4*NR *contiguous* output channels of one pixel -> wide NHWC stores.
//   stem_kernel         conv1 (3->64, 3x3 s2) + BN + ReLU straight from the caller's NCHW fp32 crops.
//   fuse_kernel         cross-resolution sum (nearest upsample folded into the read index) + ReLU.
//   head_kernel         final 1x1 conv + bias, optional heat-map write-out, per-slab arg-max.
//   decode_kernel       arg-max merge (first maximum wins) + box scaling in fp64, SimpleHRNet.py:297-308.
#include "kernels.h"
#include <stdlib.h>

namespace hrn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {  // round to nearest even
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

template <int DT>
struct Tr;
template <>
struct Tr<DT_BF16> {
    using elem = unsigned short;
    using vec = s16x8;   // 8 bf16 = 16 B = one MFMA operand
    using out4 = s16x4;  // 4 output channels
    static constexpr int KC = 32, VEC = 8;
    static __device__ __forceinline__ float ld(elem e) { return bf16_to_f32(e); }
    static __device__ __forceinline__ elem st(float f) { return f32_to_bf16(f); }
};
template <>
struct Tr<DT_F32> {
    using elem = float;
    using vec = f32x4;  // 4 fp32 = 16 B = four 16x16x4 MFMA steps
    using out4 = f32x4;
    static constexpr int KC = 16, VEC = 4;
    static __device__ __forceinline__ float ld(elem e) { return e; }
    static __device__ __forceinline__ elem st(float f) { return f; }
};

template <int DT>
__device__ __forceinline__ f32x4 mma(typename Tr<DT>::vec w, typename Tr<DT>::vec x, f32x4 acc);
template <>
__device__ __forceinline__ f32x4 mma<DT_BF16>(s16x8 w, s16x8 x, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0,
                                                   0, 0);
}
template <>
__device__ __forceinline__ f32x4 mma<DT_F32>(f32x4 w, f32x4 x, f32x4 acc) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t], x[t], acc, 0, 0, 0);
    return acc;
}

// ---------------------------------------------------------------------------------------------------
// Generic convolution.  GEMM view: D[cout][pixel] = sum_k W[cout][k] * X[k][pixel], k = tap*cin + ci.
//   block = 4 waves, wave = 16*MR pixels x 16*NR couts, grid = (ceil(m / (64*MR)), cout / (16*NR)).
//   lane (li = lane&15, g = lane>>4):  X operand = pixel li, k-group g (VEC consecutive ci of one tap);
//   W operand = packed row li, k-group g (pre-packed so the load is lane-linear: lane*16 B).
//   result: lane holds pixel li, channels ch0 + [0, 4*NR), ch0 = ng*16*NR + g*4*NR  (see pack_conv_weights).
#define GLOBAL_AS __attribute__((address_space(1)))

// PRE: the residual is fetched before the K loop instead of after it.  The 1x1 convs of layer1 have K = 64: the
// loop is two chunks long and the kernel is a chain of memory round trips (operands, residual, store); this folds
// the first two into one.  bf16, even NR only.
// WL: the weights of the block's cout group go through LDS, four K chunks at a time (LDS-DMA, double-buffered, one
// barrier per four chunks), instead of every wave fetching its own copy from L2 -- a quarter of the weight requests
// (profiles/round1_pmc_direct.txt: this kernel lives on L2 round trips).
#ifndef HRN_WL_G
#define HRN_WL_G 4
#endif
constexpr int WL_G = HRN_WL_G;
// epilogue of the generic kernel: out = [relu](acc + bias [+ residual]), zero on pad pixels, 4*NR contiguous channels per lane
template <int DT, int NR, int MR, bool PRE>
__device__ __forceinline__ void conv_direct_epilogue(const ConvArgs &p, const int ng, const int m0, const int li, const int g,
                                                     f32x4 (&acc)[MR][NR], s16x8 (&rpre)[PRE ? MR : 1][PRE ? NR / 2 : 1]) {
    using T = Tr<DT>;
    using elem = typename T::elem;
    const int ch0 = ng * 16 * NR + g * 4 * NR;
    float bias[4 * NR];
#pragma unroll
    for (int c = 0; c < 4 * NR; ++c) bias[c] = ((const GLOBAL_AS float *)p.bias)[ch0 + c];
    GLOBAL_AS elem *__restrict__ out = (GLOBAL_AS elem *)p.out;
    const GLOBAL_AS elem *__restrict__ res = (const GLOBAL_AS elem *)p.res;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int q = m0 + i * 16 + li;
        if (q >= p.m) continue;
        const int rem = q % p.out_hpwp;
        const int ho = rem / p.out_wp, wo = rem - ho * p.out_wp;
        const bool ok = (ho < p.out_h) && (wo < p.out_w);
        size_t o = (size_t)q * p.cout + ch0;
        if (p.up) {  // transposed-conv phase: scatter to (2*ho + a, 2*wo + b); pad pixels of the phase grid write nothing
            if (!ok) continue;
            o = ((size_t)(q / p.out_hpwp) * p.up_hpwp + (size_t)(2 * ho + p.up_a) * p.up_wp + 2 * wo + p.up_b) * p.cout + ch0;
        }
        if constexpr (DT == DT_BF16 && (NR % 2 == 0)) {
            // 8 contiguous channels per access: 16-byte residual loads and stores
#pragma unroll
            for (int j = 0; j < NR; j += 2) {
                s16x8 r8 = {};
                if constexpr (PRE)
                    r8 = rpre[i][j / 2];
                else if (res)
                    r8 = *(const GLOBAL_AS s16x8 *)(res + o + j * 4);
                s16x8 o8;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    float v = acc[i][j + (r >> 2)][r & 3] + bias[j * 4 + r];
                    if (PRE || res) v += T::ld((elem)r8[r]);
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (!ok) v = 0.f;
                    o8[r] = (short)T::st(v);
                }
                *(GLOBAL_AS s16x8 *)(out + o + j * 4) = o8;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                typename T::out4 r4 = {};
                if (res) r4 = *(const GLOBAL_AS typename T::out4 *)(res + o + j * 4);
                typename T::out4 o4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][j][r] + bias[j * 4 + r];
                    if (res) v += T::ld((elem)r4[r]);
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (!ok) v = 0.f;
                    o4[r] = T::st(v);
                }
                *(GLOBAL_AS typename T::out4 *)(out + o + j * 4) = o4;
            }
        }
    }
}

template <int DT, int NR, int MR, bool PRE = false, bool WL = false>
__device__ __forceinline__ void conv_direct_body(const ConvArgs &p, const int ng, const int mtile_in, char *smem = nullptr) {
    using T = Tr<DT>;
    using vec = typename T::vec;
    using elem = typename T::elem;
    typedef const GLOBAL_AS vec *gvec_p;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, g = lane >> 4;
    if (mtile_in * 64 * MR >= p.m) return;
    const int mtile = p.rev ? (p.m + 64 * MR - 1) / (64 * MR) - 1 - mtile_in : mtile_in;
    const int m0 = (mtile * 4 + wave) * (16 * MR);
    // descriptor pointers may have come from memory (grouped launch): name the address space, or every access
    // through them is a FLAT one
    const GLOBAL_AS elem *__restrict__ in = (const GLOBAL_AS elem *)p.in;

    long inrow[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int q = m0 + i * 16 + li;
        int r;
        if (q >= p.m) {
            r = 0;  // masked at the store; any mapped row will do
        } else if (p.stride == 1) {
            r = q;
        } else {
            const int n = q / p.out_hpwp, rem = q - n * p.out_hpwp;
            const int ho = rem / p.out_wp, wo = rem - ho * p.out_wp;
            r = n * p.in_hpwp + 2 * ho * p.in_wp + 2 * wo;
        }
        inrow[i] = (long)r * p.cin;
    }

    f32x4 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    s16x8 rpre[PRE ? MR : 1][PRE ? NR / 2 : 1];
    if constexpr (PRE) {
        const GLOBAL_AS elem *__restrict__ res = (const GLOBAL_AS elem *)p.res;
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; j += 2)
                rpre[i][j / 2] = *(const GLOBAL_AS s16x8 *)(res + (size_t)(m0 + i * 16 + li) * p.cout +
                                                            ng * 16 * NR + g * 4 * NR + j * 4);
    }

    int ci = g * T::VEC, tap = 0;
    while (ci >= p.cin) {
        ci -= p.cin;
        ++tap;
    }
    const int ntaps = p.ksize * p.ksize;
    const GLOBAL_AS char *__restrict__ wlane =
        (const GLOBAL_AS char *)p.w + ((size_t)ng * NR * p.kchunks * 64 + lane) * 16;

    // WL: stage chunk group gi (chunks gi*WL_G ...) of all NR fragments into buffer gi & 1: NR * WL_G pieces of 1 KiB,
    // piece t = j * WL_G + c, dealt round-robin to the four waves
    auto stage = [&](int gi) {
        const GLOBAL_AS char *wsrc = (const GLOBAL_AS char *)p.w + (size_t)ng * NR * p.kchunks * 1024;
#pragma unroll
        for (int t0 = 0; t0 < NR * WL_G; t0 += 4) {
            const int t = t0 + wave;
            const int j = t / WL_G, c = gi * WL_G + (t - j * WL_G);
            if (t < NR * WL_G && c < p.kchunks)
                __builtin_amdgcn_global_load_lds(wsrc + ((size_t)(j * p.kchunks + c) * 64 + lane) * 16,
                                                 (__attribute__((address_space(3))) void *)(smem + ((gi & 1) * NR * WL_G + t) * 1024), 16, 0, 0);
        }
    };
    if constexpr (WL) stage(0);
    // round 6, small launches (MR <= 2, bf16): a block's K loop is a chain of L2 round trips -- 27 for a 96 -> 192 stride-2 convolution, 18 us
    // per launch at 8 crops with the chip mostly idle.  The operands of KU chunks are requested together, then multiplied in the
    // same order as before (bit-identical): a quarter / half of the round trips.
    constexpr int KU = (DT == DT_BF16 && !WL && !PRE) ? (MR == 1 ? 4 : MR == 2 ? 2 : 1) : 1;
    if constexpr (KU > 1) {
        for (int kc0 = 0; kc0 < p.kchunks; kc0 += KU) {
            vec b[KU][NR], a[KU][MR];
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int kc = kc0 + u;
                if (kc < p.kchunks) {   // (wave-uniform)
#pragma unroll
                    for (int j = 0; j < NR; ++j) b[u][j] = *(gvec_p)(wlane + ((size_t)j * p.kchunks + kc) * 1024);
                    if (tap < ntaps) {
                        int tapoff = 0;
                        if (p.ksize == 3) {
                            const int dh = (tap * 11) >> 5, dw = tap - dh * 3;
                            tapoff = (dh - 1) * p.in_wp + (dw - 1);
                        } else if (p.ksize == 2) {
                            tapoff = tap == 0 ? p.taps[0] : tap == 1 ? p.taps[1] : tap == 2 ? p.taps[2] : p.taps[3];
                        }
                        const long aoff = (long)tapoff * p.cin + ci;
#pragma unroll
                        for (int i = 0; i < MR; ++i) a[u][i] = *(gvec_p)(in + inrow[i] + aoff);
                    } else {
#pragma unroll
                        for (int i = 0; i < MR; ++i) a[u][i] = vec{};
                    }
                    ci += T::KC;
                    while (ci >= p.cin) {
                        ci -= p.cin;
                        ++tap;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < KU; ++u)
                if (kc0 + u < p.kchunks) {
#pragma unroll
                    for (int i = 0; i < MR; ++i)
#pragma unroll
                        for (int j = 0; j < NR; ++j) acc[i][j] = mma<DT>(b[u][j], a[u][i], acc[i][j]);
                }
        }
    } else
    for (int kc = 0; kc < p.kchunks; ++kc) {
        vec b[NR];
        if constexpr (WL) {
            const int gi = kc / WL_G, c = kc - gi * WL_G;
            if (c == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();  // group gi has landed for everybody; everybody is done with the other buffer
                if ((gi + 1) * WL_G < p.kchunks) stage(gi + 1);
            }
            const char *wl = smem + ((gi & 1) * NR * WL_G + c) * 1024 + lane * 16;
#pragma unroll
            for (int j = 0; j < NR; ++j) b[j] = *(const vec *)(wl + j * WL_G * 1024);
        } else {
#pragma unroll
            for (int j = 0; j < NR; ++j) b[j] = *(gvec_p)(wlane + ((size_t)j * p.kchunks + kc) * 1024);
        }
        vec a[MR];
        if (tap < ntaps) {
            int tapoff = 0;
            if (p.ksize == 3) {
                const int dh = (tap * 11) >> 5, dw = tap - dh * 3;
                tapoff = (dh - 1) * p.in_wp + (dw - 1);
            } else if (p.ksize == 2) {  // the four live taps of a transposed-conv phase
                tapoff = tap == 0 ? p.taps[0] : tap == 1 ? p.taps[1] : tap == 2 ? p.taps[2] : p.taps[3];
            }
            const long aoff = (long)tapoff * p.cin + ci;
#pragma unroll
            for (int i = 0; i < MR; ++i) a[i] = *(gvec_p)(in + inrow[i] + aoff);
        } else {  // K padding: weights are zero there, feed zeros
#pragma unroll
            for (int i = 0; i < MR; ++i) a[i] = vec{};
        }
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j) acc[i][j] = mma<DT>(b[j], a[i], acc[i][j]);
        ci += T::KC;
        while (ci >= p.cin) {
            ci -= p.cin;
            ++tap;
        }
    }

    conv_direct_epilogue<DT, NR, MR, PRE>(p, ng, m0, li, g, acc, rpre);
}

// XL (round 6): the stride-2 3x3 convolutions with cin % 32 == 0 (96 / 192 / 256 / 384 input channels: 1.5 ms of a 256-crop W48 pass on
// this kernel at 240-470 TFLOP/s, matrix pipe 19 % busy, waves parked on the L2 round trip of every tap's pixel fragments).  Same tiles,
// same weight image, same K order and MFMA sequence as conv_direct_body<bf16, NR, 4, false, WL> -- results are bit-identical -- but the
// pixel fragments of chunk kc + 2 are REQUESTED (LDS-DMA, per-lane source = the lane's 16 bytes of its pixel, lane-linear destination = the
// fragment image itself) while chunk kc is multiplied: the prefetch buffer is LDS (a private ring of XL_D slots per wave, no barrier), not
// registers -- the register-ring version cost a wave per SIMD (EXPERIMENTS, round 1).  Weights: XL_G chunks per block-wide stage, double
// buffered, one barrier per stage.  All waits on vector memory are counted (in order): 8 operations behind a weight stage, 11 behind a
// chunk's pixels.  LDS: 2 * XL_G * NR KiB + 4 waves * XL_D * 4 KiB = 72 KiB for NR = 6: two blocks per CU.
constexpr int XL_G = 2, XL_D = 3;
constexpr int xl_lds_bytes(int nr) { return 2 * XL_G * nr * 1024 + 4 * XL_D * 4 * 1024; }
__device__ __forceinline__ bool conv_xl_ok(const ConvArgs &p) {
    return p.stride == 2 && p.ksize == 3 && (p.cin & 31) == 0 && !p.up && !p.res && p.kchunks >= 2 * XL_G;
}
__device__ __forceinline__ void xl_glds(const GLOBAL_AS char *base, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" ::"v"(voff), "s"(lds_dst), "s"(base) : "memory", "m0");
}
template <int NR>
__device__ __forceinline__ void conv_direct_xl_body(const ConvArgs &p, const int ng, const int mtile_in, char *smem) {
    constexpr int MR = 4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, g = lane >> 4;
    if (mtile_in * 64 * MR >= p.m) return;   // (block-uniform: ahead of every barrier)
    const int mtile = p.rev ? (p.m + 64 * MR - 1) / (64 * MR) - 1 - mtile_in : mtile_in;
    const int m0 = (mtile * 4 + wave) * (16 * MR);
    const int cin = p.cin, in_wp = p.in_wp, kchunks = p.kchunks;
    const int spt = cin >> 5;                       // 32-channel slices (= K chunks) per tap
    // scalar base = the tensor's row -(in_wp + 1) (the front guard rows: tap (-1, -1) of row 0 is a valid address), per-lane offsets from it
    const int guard = (in_wp + 1) * cin;
    const GLOBAL_AS char *const base = (const GLOBAL_AS char *)p.in - (size_t)guard * 2;
    unsigned xoff[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int q = m0 + i * 16 + li;
        int r = 0;   // (rows past the end: any mapped row; masked at the store)
        if (q < p.m) {
            const int n = q / p.out_hpwp, rem = q - n * p.out_hpwp;
            const int ho = rem / p.out_wp, wo = rem - ho * p.out_wp;
            r = n * p.in_hpwp + 2 * ho * in_wp + 2 * wo;
        }
        xoff[i] = ((unsigned)r * (unsigned)cin + (unsigned)(guard + g * 8)) * 2u;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned xring = lds0 + 2 * XL_G * NR * 1024 + wave * (XL_D * MR * 1024);
    const GLOBAL_AS char *const wsrc = (const GLOBAL_AS char *)p.w + (size_t)ng * NR * kchunks * 1024;
    const unsigned lane16 = lane * 16;
    // pixel fragments of chunk kc -> ring slot kc % XL_D (chunks past the end: the last chunk again, into a slot nobody reads any more --
    // every wave issues the same number of operations per chunk, which is what the counted waits rely on)
    auto px_issue = [&](int kc) {
        const int slot = kc % XL_D;
        const int kq = kc < kchunks ? kc : kchunks - 1;
        const int tap = kq / spt, sl = kq - tap * spt;
        const int dh = (tap * 11) >> 5, dw = tap - dh * 3;
        const GLOBAL_AS char *sb = base + ((long)((dh - 1) * in_wp + (dw - 1)) * cin + sl * 32) * 2;
#pragma unroll
        for (int i = 0; i < MR; ++i) xl_glds(sb, xoff[i], xring + (slot * MR + i) * 1024);
    };
    // weight stage gi (chunks gi * XL_G ...) -> buffer gi & 1: NR * XL_G pieces of 1 KiB, three per wave
    auto w_issue = [&](int gi) {
#pragma unroll
        for (int u = 0; u < (NR * XL_G + 3) / 4; ++u) {
            int t = wave + 4 * u;
            if (t >= NR * XL_G) t = NR * XL_G - 1;   // (NR * XL_G = 12 = 3 per wave exactly for NR = 6)
            const int j = t / XL_G;
            int c = gi * XL_G + (t - j * XL_G);
            if (c >= kchunks) c = kchunks - 1;
            xl_glds(wsrc + (size_t)(j * kchunks + c) * 1024, lane16, lds0 + ((gi & 1) * NR * XL_G + t) * 1024);
        }
    };
    static_assert((NR * XL_G) % 4 == 0, "weight pieces per wave");
    constexpr int WOPS = NR * XL_G / 4;   // 3
    f32x4 acc[MR][NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    w_issue(0);
    px_issue(0);
    px_issue(1);
    for (int kc = 0; kc < kchunks; ++kc) {
        const int gi = kc / XL_G, c = kc - gi * XL_G;
        if (c == 0) {
            // this wave's pieces of stage gi have landed (behind them: the pixels of two chunks), then everybody's; everybody is also
            // done reading the other buffer
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"((XL_D - 1) * MR) : "memory");
            __builtin_amdgcn_s_barrier();
            w_issue(gi + 1);
        }
        px_issue(kc + XL_D - 1);
        // the pixels of chunk kc: behind them the other XL_D - 2 chunks in flight, a weight stage, and the chunk just requested
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"((XL_D - 1) * MR + WOPS) : "memory");
        s16x8 a[MR], b[NR];
        const unsigned xa = xring + (kc % XL_D) * (MR * 1024) + lane16;
        const unsigned wa = lds0 + ((gi & 1) * NR * XL_G + c) * 1024 + lane16;
#pragma unroll
        for (int i = 0; i < MR; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[i]) : "v"(xa), "i"(i * 1024));
#pragma unroll
        for (int j = 0; j < NR; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[j]) : "v"(wa), "i"(j * XL_G * 1024));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j) acc[i][j] = mma<DT_BF16>(b[j], a[i], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the dummy requests past the end: nothing of this block may land in LDS after it)
    s16x8 rpre[1][1];
    conv_direct_epilogue<DT_BF16, NR, MR, false>(p, ng, m0, li, g, acc, rpre);
}

// one convolution per launch.  1-D grid, cout tile fastest: the blocks that share an activation tile are dispatched
// together, so the tile is fetched from HBM once and re-read from L2.  The hardware places block b on XCD b % 8
// (private L2s): the cout tiles of one M tile are issued 8 ids apart.
template <int DT, int NR, int MR, bool PRE, bool WL = false>
__global__ __launch_bounds__(256) void conv_direct_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_direct[];
    const int ngroups = p.cout / (16 * NR);
    const int ng = (blockIdx.x >> 3) % ngroups;
    const int mtile = (blockIdx.x / (8 * ngroups)) * 8 + (blockIdx.x & 7);
    if constexpr (WL && DT == DT_BF16 && NR == 6 && MR == 4 && !PRE) {
        if (p.xlds && conv_xl_ok(p)) {
            conv_direct_xl_body<NR>(p, ng, mtile, smem_direct);
            return;
        }
    }
    conv_direct_body<DT, NR, MR, PRE, WL>(p, ng, mtile, smem_direct);
}

// several independent convolutions per launch: block b runs map[b] = (problem | cout tile << 8, M tile) of the
// device-resident descriptor array (the host orders the map so that convolutions reading the same tensor sit on
// the same XCD at the same time, hrnet_mi355.cpp: direct_group_blocks)
template <int DT, int NR, int MR, bool WL = false>
__global__ __launch_bounds__(256) void conv_direct_group_kernel(const ConvArgs *__restrict__ probs,
                                                                const int2 *__restrict__ map) {
    extern __shared__ __attribute__((aligned(16))) char smem_direct[];
    const int2 e = map[blockIdx.x];
    const int prob = __builtin_amdgcn_readfirstlane(e.x & 255), ng = __builtin_amdgcn_readfirstlane(e.x >> 8);
    const int mtile = __builtin_amdgcn_readfirstlane(e.y);
    const ConvArgs p = probs[prob];
    if constexpr (WL && DT == DT_BF16 && NR == 6 && MR == 4) {
        if (p.xlds && conv_xl_ok(p)) {
            conv_direct_xl_body<NR>(p, ng, mtile, smem_direct);
            return;
        }
    }
    conv_direct_body<DT, NR, MR, false, WL>(p, ng, mtile, smem_direct);
}

// dynamic LDS of the WL instantiations: the weight double buffer; wlds == 2 (a launch with XL members): room for the XL layout too
template <int DT, int NR>
static constexpr int direct_lds_bytes(int wlds) {
    const int wl = 2 * WL_G * NR * 1024;
    if (DT == DT_BF16 && NR == 6 && wlds == 2) return wl > xl_lds_bytes(NR) ? wl : xl_lds_bytes(NR);
    return wl;
}

template <int DT, int NR>
static hipError_t launch_conv_group_t(const ConvArgs *probs, const int2 *map, int nblocks, int mr, int wlds, hipStream_t s) {
    if (mr == 1)
        hipLaunchKernelGGL((conv_direct_group_kernel<DT, NR, 1>), dim3(nblocks), dim3(256), 0, s, probs, map);
    else if (mr == 2)
        hipLaunchKernelGGL((conv_direct_group_kernel<DT, NR, 2>), dim3(nblocks), dim3(256), 0, s, probs, map);
    else if (DT == DT_BF16 && wlds)
    {
        const int lds = direct_lds_bytes<DT, NR>(wlds);
        static std::atomic<unsigned long long> lds_set{0};   // (more than 64 KiB of dynamic LDS: per device, kernels.h set_dynamic_lds)
        if (lds > 65536) {
            const hipError_t e = set_dynamic_lds((const void *)conv_direct_group_kernel<DT, NR, 4, DT == DT_BF16>, direct_lds_bytes<DT, NR>(2), lds_set);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((conv_direct_group_kernel<DT, NR, 4, DT == DT_BF16>), dim3(nblocks), dim3(256), lds, s, probs, map);
    }
    else
        hipLaunchKernelGGL((conv_direct_group_kernel<DT, NR, 4>), dim3(nblocks), dim3(256), 0, s, probs, map);
    return hipGetLastError();
}

// mr = 16-pixel fragments per wave (4, 2 or 1): the host shrinks the M tile when a launch would not fill the chip
hipError_t launch_conv_group(int dtype, const ConvArgs *probs_dev, const void *map_dev, int nblocks, int nr, int mr, int wlds,
                             hipStream_t s) {
    if (nblocks <= 0) return hipSuccess;
    const int2 *map = (const int2 *)map_dev;
    if (dtype == DT_BF16) {
        if (nr == 6) return launch_conv_group_t<DT_BF16, 6>(probs_dev, map, nblocks, mr, wlds, s);
        if (nr == 4) return launch_conv_group_t<DT_BF16, 4>(probs_dev, map, nblocks, mr, wlds, s);
        if (nr == 3) return launch_conv_group_t<DT_BF16, 3>(probs_dev, map, nblocks, mr, wlds, s);
        if (nr == 2) return launch_conv_group_t<DT_BF16, 2>(probs_dev, map, nblocks, mr, wlds, s);
    } else {
        if (nr == 4) return launch_conv_group_t<DT_F32, 4>(probs_dev, map, nblocks, mr, wlds, s);
        if (nr == 3) return launch_conv_group_t<DT_F32, 3>(probs_dev, map, nblocks, mr, wlds, s);
        if (nr == 2) return launch_conv_group_t<DT_F32, 2>(probs_dev, map, nblocks, mr, wlds, s);
    }
    return hipErrorInvalidValue;
}

template <int DT, int NR, int MR, bool PRE>
static hipError_t launch_conv_tt(const ConvArgs &a, hipStream_t s) {
    const int mtiles = (a.m + 64 * MR - 1) / (64 * MR);
    dim3 grid(((mtiles + 7) / 8) * 8 * (a.cout / (16 * NR)));
    hipLaunchKernelGGL((conv_direct_kernel<DT, NR, MR, PRE>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

template <int DT, int NR>
static hipError_t launch_conv_t(const ConvArgs &a, hipStream_t s) {
    if constexpr (DT == DT_BF16 && NR % 2 == 0) {
        const int pre_mode = a.pre_mode;
        if (a.res && a.ksize == 1 && pre_mode == 1) return launch_conv_tt<DT, NR, 4, true>(a, s);
        if (a.res && a.ksize == 1 && pre_mode == 2) return launch_conv_tt<DT, NR, 2, true>(a, s);
        if (a.res && a.ksize == 1 && pre_mode == 3) return launch_conv_tt<DT, NR, 2, false>(a, s);
    }
    // small launches (few crops): shorter M tiles, so that the chip fills and a block's serial K loop shrinks with it
    const int ngroups = a.cout / (16 * NR);
    const long blocks4 = (long)((a.m + 255) / 256) * ngroups;
    if (blocks4 < 256) return launch_conv_tt<DT, NR, 1, false>(a, s);
    if (blocks4 < 512) return launch_conv_tt<DT, NR, 2, false>(a, s);
    if constexpr (DT == DT_BF16)
        if (a.wlds && a.kchunks >= 2 * WL_G) {
            const int mtiles = (a.m + 255) / 256;
            dim3 grid(((mtiles + 7) / 8) * 8 * (a.cout / (16 * NR)));
            const int lds = direct_lds_bytes<DT, NR>(a.xlds ? 2 : 1);
            static std::atomic<unsigned long long> lds_set{0};
            if (lds > 65536) {
                const hipError_t e = set_dynamic_lds((const void *)conv_direct_kernel<DT, NR, 4, false, true>, direct_lds_bytes<DT, NR>(2), lds_set);
                if (e != hipSuccess) return e;
            }
            hipLaunchKernelGGL((conv_direct_kernel<DT, NR, 4, false, true>), grid, dim3(256), lds, s, a);
            return hipGetLastError();
        }
    return launch_conv_tt<DT, NR, 4, false>(a, s);
}

hipError_t launch_conv(int dtype, const ConvArgs &a, int nr, hipStream_t s) {
    if (a.m <= 0) return hipSuccess;
    if (dtype == DT_BF16) {
        if (nr == 6) return launch_conv_t<DT_BF16, 6>(a, s);
        if (nr == 4) return launch_conv_t<DT_BF16, 4>(a, s);
        if (nr == 3) return launch_conv_t<DT_BF16, 3>(a, s);
        if (nr == 2) return launch_conv_t<DT_BF16, 2>(a, s);
    } else {
        if (nr == 4) return launch_conv_t<DT_F32, 4>(a, s);
        if (nr == 3) return launch_conv_t<DT_F32, 3>(a, s);
        if (nr == 2) return launch_conv_t<DT_F32, 2>(a, s);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------
// Stem conv1: 3->64, 3x3, stride 2, pad 1 (+folded BN, ReLU).  Input is the caller's NCHW fp32 batch,
// arithmetic is fp32 in both modes.  blockIdx.y = group of 16 output channels (weights are block-uniform
// -> scalar loads), threadIdx -> output row q of the flat padded layout.
template <int DT>
__global__ __launch_bounds__(256) void stem_kernel(const StemArgs p) {
    using T = Tr<DT>;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int m = p.n * p.out_hpwp;
    if (q >= m) return;
    const int n = q / p.out_hpwp, rem = q - n * p.out_hpwp;
    const int ho = rem / p.out_wp, wo = rem - ho * p.out_wp;
    const bool ok = ho < p.out_h && wo < p.out_w;
    // one thread = one output pixel x all 64 channels: the 27 inputs are fetched once, the 128-byte (bf16)
    // NHWC row is written with full 16-byte stores; weights are wave-uniform -> scalar loads / SGPR operands
    float x[27];
    const float *img = p.images + (size_t)n * 3 * p.H * p.W;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = 2 * ho + kh - 1;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ix = 2 * wo + kw - 1;
                float v = 0.f;
                if (ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                    v = img[((size_t)ci * p.H + iy) * p.W + (p.flip ? p.W - 1 - ix : ix)];  // flip: the mirrored crop (flip-TTA)
                x[ci * 9 + kh * 3 + kw] = v;
            }
        }
    typename T::elem *o = (typename T::elem *)p.out + (size_t)q * 64;
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) {
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = p.bias[cg * 16 + c];
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const float *wk = p.w + k * 64 + cg * 16;
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] = fmaf(x[k], wk[c], acc[c]);
        }
#pragma unroll
        for (int v = 0; v < 16 / T::VEC; ++v) {
            typename T::vec ov;
#pragma unroll
            for (int r = 0; r < T::VEC; ++r) {
                const float f = ok ? fmaxf(acc[v * T::VEC + r], 0.f) : 0.f;
                ov[r] = T::st(f);
            }
            *(typename T::vec *)(o + cg * 16 + v * T::VEC) = ov;
        }
    }
}

// PoseResNet conv1: 3->64, 7x7, stride 2, pad 3 (+folded BN, ReLU) from the caller's NCHW fp32 batch.  fp32 fma chain in
// both modes (k = (ci, kh, kw) ascending, starting at the bias); one thread = one output pixel, 16 channels at a time
// (the weights are wave-uniform: scalar loads), the 147 inputs re-read from L1 for each of the four channel groups.
template <int DT>
__global__ __launch_bounds__(256) void stem7_kernel(const Stem7Args p) {
    using T = Tr<DT>;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int m = p.n * p.out_hpwp;
    if (q >= m) return;
    const int n = q / p.out_hpwp, rem = q - n * p.out_hpwp;
    const int ho = rem / p.out_wp, wo = rem - ho * p.out_wp;
    const bool ok = ho < p.out_h && wo < p.out_w;
    const float *img = p.images + (size_t)n * 3 * p.H * p.W;
    typename T::elem *o = (typename T::elem *)p.out + (size_t)q * 64;
    for (int cg = 0; cg < 4; ++cg) {
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = p.bias[cg * 16 + c];
        if (ok) {
            for (int ci = 0; ci < 3; ++ci)
                for (int kh = 0; kh < 7; ++kh) {
                    const int iy = 2 * ho + kh - 3;
                    if (iy < 0 || iy >= p.H) continue;  // zero padding: the term vanishes
#pragma unroll
                    for (int kw = 0; kw < 7; ++kw) {
                        const int ix = 2 * wo + kw - 3;
                        if (ix < 0 || ix >= p.W) continue;
                        const float x = img[((size_t)ci * p.H + iy) * p.W + (p.flip ? p.W - 1 - ix : ix)];
                        const float *wk = p.w + ((ci * 7 + kh) * 7 + kw) * 64 + cg * 16;
#pragma unroll
                        for (int c = 0; c < 16; ++c) acc[c] = fmaf(x, wk[c], acc[c]);
                    }
                }
        }
#pragma unroll
        for (int v = 0; v < 16 / T::VEC; ++v) {
            typename T::vec ov;
#pragma unroll
            for (int r = 0; r < T::VEC; ++r) ov[r] = T::st(ok ? fmaxf(acc[v * T::VEC + r], 0.f) : 0.f);
            *(typename T::vec *)(o + cg * 16 + v * T::VEC) = ov;
        }
    }
}

// bf16 mode: the 7x7 stem on MFMA.  K = 147 (ci, kh, kw) padded to five 32-wide chunks; a lane gathers its 8 k-values
// of one output pixel per chunk straight from the NCHW fp32 crop (rounded to bf16), accumulators start at the bias.
__global__ __launch_bounds__(256) void stem7_mfma_kernel(const Stem7Args p) {
    constexpr int MR = 4, NR = 4, KCH = 5;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int m = p.n * p.out_hpwp;
    const int q0 = (blockIdx.x * 4 + wave) * 64;
    if (q0 >= m) return;
    f32x4 acc[MR][NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const f32x4 b = *(const f32x4 *)(p.bias + g * 4 * NR + j * 4);
#pragma unroll
        for (int i = 0; i < MR; ++i) acc[i][j] = b;
    }
    bool okp[MR];
    int hy[MR], wx[MR];
    const float *img[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int q = q0 + i * 16 + li;
        const int n = q / p.out_hpwp, rem = q - n * p.out_hpwp;
        const int ho = rem / p.out_wp, wo = rem - ho * p.out_wp;
        okp[i] = q < m && ho < p.out_h && wo < p.out_w;
        hy[i] = 2 * ho - 3, wx[i] = 2 * wo - 3;
        img[i] = p.images + (size_t)(q < m ? n : 0) * 3 * p.H * p.W;
    }
    for (int kc = 0; kc < KCH; ++kc) {
        s16x8 wf[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) wf[j] = *(const s16x8 *)((const char *)p.wp + ((kc * NR + j) * 64 + lane) * 16);
        int dci[8], dkh[8], dkw[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = kc * 32 + g * 8 + e;
            dci[e] = k / 49, dkh[e] = (k % 49) / 7, dkw[e] = k % 7;
        }
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            s16x8 xf;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int iy = hy[i] + dkh[e], ix = wx[i] + dkw[e];
                float v = 0.f;
                if (okp[i] && kc * 32 + g * 8 + e < 147 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                    v = img[i][((size_t)dci[e] * p.H + iy) * p.W + (p.flip ? p.W - 1 - ix : ix)];
                xf[e] = (short)f32_to_bf16(v);
            }
#pragma unroll
            for (int j = 0; j < NR; ++j) acc[i][j] = mma<DT_BF16>(wf[j], xf, acc[i][j]);
        }
    }
    unsigned short *out = (unsigned short *)p.out;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int q = q0 + i * 16 + li;
        if (q >= m) continue;
        unsigned short *o = out + (size_t)q * 64 + g * 4 * NR;
#pragma unroll
        for (int j = 0; j < NR; j += 2) {
            s16x8 o8;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float v = okp[i] ? fmaxf(acc[i][j + (r >> 2)][r & 3], 0.f) : 0.f;
                o8[r] = (short)f32_to_bf16(v);
            }
            *(s16x8 *)(o + j * 4) = o8;
        }
    }
}

hipError_t launch_stem7(int dtype, const Stem7Args &a, hipStream_t s) {
    const int m = a.n * a.out_hpwp;
    if (m <= 0) return hipSuccess;
    dim3 grid((m + 255) / 256);
    if (dtype == DT_BF16 && a.wp)
        hipLaunchKernelGGL(stem7_mfma_kernel, grid, dim3(256), 0, s, a);
    else if (dtype == DT_BF16)
        hipLaunchKernelGGL(stem7_kernel<DT_BF16>, grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(stem7_kernel<DT_F32>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// MaxPool2d(kernel 3, stride 2, padding 1) on the flat padded layout.  Its input is post-ReLU (>= 0) and the layout's
// pad pixels are zeros, so a zero pad gives the same maximum as PyTorch's implicit -inf pad.
template <int DT>
__global__ __launch_bounds__(256) void maxpool_kernel(const PoolArgs p) {
    using T = Tr<DT>;
    using vec = typename T::vec;
    const int cvn = p.c / T::VEC;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)p.n * p.out_hpwp * cvn;
    if (idx >= total) return;
    const int q = (int)(idx / cvn), cv = (int)(idx - (long)q * cvn);
    const int n = q / p.out_hpwp, rem = q - n * p.out_hpwp;
    const int ho = rem / p.out_wp, wo = rem - ho * p.out_wp;
    vec o = {};
    if (ho < p.out_h && wo < p.out_w) {
        float best[T::VEC];
#pragma unroll
        for (int e = 0; e < T::VEC; ++e) best[e] = 0.f;   // >= every pad; inputs are >= 0
        const typename T::elem *base = (const typename T::elem *)p.in + (size_t)cv * T::VEC;
#pragma unroll
        for (int dh = -1; dh <= 1; ++dh)
#pragma unroll
            for (int dw = -1; dw <= 1; ++dw) {
                // row -1 / column -1 are pad (or guard) rows of the flat layout: zeros
                const long row = (long)n * p.in_hpwp + (long)(2 * ho + dh) * p.in_wp + (2 * wo + dw);
                const vec v = *(const vec *)(base + row * p.c);
#pragma unroll
                for (int e = 0; e < T::VEC; ++e) best[e] = fmaxf(best[e], T::ld((typename T::elem)v[e]));
            }
#pragma unroll
        for (int e = 0; e < T::VEC; ++e) o[e] = T::st(best[e]);
    }
    *(vec *)((typename T::elem *)p.out + (size_t)q * p.c + (size_t)cv * T::VEC) = o;
}

hipError_t launch_maxpool(int dtype, const PoolArgs &a, hipStream_t s) {
    const long total = (long)a.n * a.out_hpwp * (a.c / (dtype == DT_BF16 ? 8 : 4));
    if (total <= 0) return hipSuccess;
    dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == DT_BF16)
        hipLaunchKernelGGL(maxpool_kernel<DT_BF16>, grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(maxpool_kernel<DT_F32>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// bf16 mode: the same 3->64 stride-2 convolution on MFMA.  K = 27 (ci, kh, kw) padded to one 32-wide chunk; a
// lane gathers its 8 k-values of one output pixel straight from the NCHW fp32 crop (rounded to bf16 -- every
// later activation is bf16 as well), the 4 KiB weight image is held in registers, and one wave turns 64 pixels x
// 64 channels with 16 MFMAs instead of 1728 scalar-weight FMAs per pixel.  Output: 32 contiguous bytes per lane.
__global__ __launch_bounds__(256) void stem_mfma_kernel(const StemArgs p) {
    constexpr int MR = 4, NR = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int m = p.n * p.out_hpwp;
    const int q0 = (blockIdx.x * 4 + wave) * 64;
    if (q0 >= m) return;
    s16x8 wf[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) wf[j] = *(const s16x8 *)((const char *)p.wp + (j * 64 + lane) * 16);
    f32x4 acc[MR][NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const f32x4 b = *(const f32x4 *)(p.bias + g * 4 * NR + j * 4);
#pragma unroll
        for (int i = 0; i < MR; ++i) acc[i][j] = b;
    }
    // this lane's 8 taps: k = g*8 + e -> (ci, kh, kw); k >= 27 is zero padding
    int dci[8], dkh[8], dkw[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = g * 8 + e;
        dci[e] = k / 9, dkh[e] = (k % 9) / 3 - 1, dkw[e] = k % 3 - 1;
    }
    bool okp[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int q = q0 + i * 16 + li;
        const int n = q / p.out_hpwp, rem = q - n * p.out_hpwp;
        const int ho = rem / p.out_wp, wo = rem - ho * p.out_wp;
        okp[i] = q < m && ho < p.out_h && wo < p.out_w;
        const float *img = p.images + (size_t)n * 3 * p.H * p.W;
        s16x8 xf;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int iy = 2 * ho + dkh[e], ix = 2 * wo + dkw[e];
            float v = 0.f;
            if (okp[i] && g * 8 + e < 27 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                v = img[((size_t)dci[e] * p.H + iy) * p.W + (p.flip ? p.W - 1 - ix : ix)];
            xf[e] = (short)f32_to_bf16(v);
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = mma<DT_BF16>(wf[j], xf, acc[i][j]);
    }
    unsigned short *out = (unsigned short *)p.out;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int q = q0 + i * 16 + li;
        if (q >= m) continue;
        unsigned short *o = out + (size_t)q * 64 + g * 4 * NR;
#pragma unroll
        for (int j = 0; j < NR; j += 2) {
            s16x8 o8;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float v = okp[i] ? fmaxf(acc[i][j + (r >> 2)][r & 3], 0.f) : 0.f;
                o8[r] = (short)f32_to_bf16(v);
            }
            *(s16x8 *)(o + j * 4) = o8;
        }
    }
}

hipError_t launch_stem(int dtype, const StemArgs &a, hipStream_t s) {
    const int m = a.n * a.out_hpwp;
    if (m <= 0) return hipSuccess;
    dim3 grid((m + 255) / 256);
    if (dtype == DT_BF16 && a.wp)
        hipLaunchKernelGGL(stem_mfma_kernel, grid, dim3(256), 0, s, a);
    else if (dtype == DT_BF16)
        hipLaunchKernelGGL(stem_kernel<DT_BF16>, grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(stem_kernel<DT_F32>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Cross-resolution fuse (hrnet.py:60-69): out = relu(t0 + t1 + ...), left to right, fp32.
// A term with shift s is a lower-resolution tensor read at (r>>s, c>>s) (nn.Upsample nearest, integer
// scale).  Pad pixels map to pad pixels, so zeros propagate without a mask.
template <int DT>
__device__ __forceinline__ void fuse_body(const FuseArgs &p, const long block, const long nblocks) {
    using T = Tr<DT>;
    using vec = typename T::vec;
    const int cvn = p.c / T::VEC;
    const long bid = p.rev ? nblocks - 1 - block : block;
    const long idx = bid * 256 + threadIdx.x;
    const long total = (long)p.m * cvn;
    if (idx >= total) return;
    const int q = (int)(idx / cvn), cv = (int)(idx - (long)q * cvn);
    const int n = q / p.hpwp, rem = q - n * p.hpwp;
    const int r = rem / p.wp, c = rem - r * p.wp;
    float sum[T::VEC];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t >= p.nterms) break;
        const FuseTerm ft = p.t[t];
        const long row = ft.shift == 0 ? (long)q : (long)n * ft.hpwp + (long)(r >> ft.shift) * ft.wp + (c >> ft.shift);
        const vec v = *(const vec *)((const typename T::elem *)ft.ptr + row * p.c + cv * T::VEC);
#pragma unroll
        for (int e = 0; e < T::VEC; ++e) {
            const float f = T::ld((typename T::elem)v[e]);
            sum[e] = (t == 0) ? f : sum[e] + f;
        }
    }
    vec o;
#pragma unroll
    for (int e = 0; e < T::VEC; ++e) o[e] = T::st(fmaxf(sum[e], 0.f));
    *(vec *)((typename T::elem *)p.out + (long)q * p.c + cv * T::VEC) = o;
}

template <int DT>
__global__ __launch_bounds__(256) void fuse_kernel(const FuseArgs p) {
    fuse_body<DT>(p, blockIdx.x, gridDim.x);
}

// all outputs of one StageModule's fuse (hrnet.py:60-69) in ONE launch: they are independent, and as separate launches each
// of the three or four HBM-bound kernels pays its own ramp and drain (23 launches per pass -> 8)
template <int DT>
__global__ __launch_bounds__(256) void fuse_group_kernel(const FuseGroupArgs g) {
    int k = 0;
    long first = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (k == i && i + 1 < g.nf && (long)blockIdx.x >= g.block_end[i]) first = g.block_end[i], k = i + 1;
    // (a by-value kernel argument indexed with a runtime k would live in scratch: pick with wave-uniform branches)
    if (k == 0)
        fuse_body<DT>(g.f[0], (long)blockIdx.x - first, (long)g.block_end[0] - first);
    else if (k == 1)
        fuse_body<DT>(g.f[1], (long)blockIdx.x - first, (long)g.block_end[1] - first);
    else if (k == 2)
        fuse_body<DT>(g.f[2], (long)blockIdx.x - first, (long)g.block_end[2] - first);
    else
        fuse_body<DT>(g.f[3], (long)blockIdx.x - first, (long)g.block_end[3] - first);
}

hipError_t launch_fuse(int dtype, const FuseArgs &a, hipStream_t s) {
    const long total = (long)a.m * (a.c / (dtype == DT_BF16 ? 8 : 4));
    if (total <= 0) return hipSuccess;
    dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == DT_BF16)
        hipLaunchKernelGGL(fuse_kernel<DT_BF16>, grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(fuse_kernel<DT_F32>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_fuse_group(int dtype, FuseGroupArgs &g, hipStream_t s) {
    long end = 0;
    for (int i = 0; i < g.nf; ++i) {
        const long total = (long)g.f[i].m * (g.f[i].c / (dtype == DT_BF16 ? 8 : 4));
        end += (total + 255) / 256;
        g.block_end[i] = (int)end;
    }
    if (end <= 0) return hipSuccess;
    if (dtype == DT_BF16)
        hipLaunchKernelGGL(fuse_group_kernel<DT_BF16>, dim3((unsigned)end), dim3(256), 0, s, g);
    else
        hipLaunchKernelGGL(fuse_group_kernel<DT_F32>, dim3((unsigned)end), dim3(256), 0, s, g);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Head: final_layer 1x1 conv c->joints WITH bias (hrnet.py:155,187), fp32 weights/accumulate.
// grid = (slabs, n); each block scans slab_px pixels of one crop, optionally writes the NCHW fp32
// heat-maps, and leaves one (max, first index) candidate per joint.
constexpr int kMaxJoints = 32;

// Arg-max order of np.argmax / torch.max (SimpleHRNet.py:300): the first maximum wins and a NaN is a maximum (numpy
// returns the index of the first NaN).  `kNoIdx` marks "nothing seen yet": any real candidate beats it, so a map of
// -inf everywhere decodes to index 0 like numpy, not to the sentinel.
constexpr int kNoIdx = 0x7fffffff;
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn != bn) return vn;
    return v > bv || ((v == bv || vn) && i < bi);
}
// scan step for candidates visited in increasing index order
__device__ __forceinline__ bool takes(float v, float bv, int bi) { return v > bv || bi == kNoIdx || (v != v && bv == bv); }

template <int DT>
__global__ __launch_bounds__(256) void head_kernel(const HeadArgs p) {
    using T = Tr<DT>;
    using vec = typename T::vec;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *wsh = smem;                       // [joints][c]
    float *red_v = smem + p.joints * p.c;    // [4][joints]
    int *red_i = (int *)(red_v + 4 * kMaxJoints);
    for (int i = threadIdx.x; i < p.joints * p.c; i += 256) wsh[i] = p.wgt[i];
    __syncthreads();
    const int slab = blockIdx.x, n = blockIdx.y;
    const int hw = p.h * p.w;
    const int px_end = min(hw, (slab + 1) * p.slab_px);
    float bv[kMaxJoints];
    int bi[kMaxJoints];
#pragma unroll
    for (int j = 0; j < kMaxJoints; ++j) {
        bv[j] = -INFINITY;
        bi[j] = kNoIdx;
    }
    for (int px = slab * p.slab_px + threadIdx.x; px < px_end; px += 256) {
        const int r = px / p.w, c = px - r * p.w;
        const typename T::elem *row = (const typename T::elem *)p.in + ((size_t)n * p.hpwp + r * p.wp + c) * p.c;
        float acc[kMaxJoints];
#pragma unroll
        for (int j = 0; j < kMaxJoints; ++j) acc[j] = 0.f;
        for (int c0 = 0; c0 < p.c; c0 += T::VEC) {
            const vec v = *(const vec *)(row + c0);
#pragma unroll
            for (int e = 0; e < T::VEC; ++e) {
                const float x = T::ld((typename T::elem)v[e]);
#pragma unroll
                for (int j = 0; j < kMaxJoints; ++j)
                    if (j < p.joints) acc[j] = fmaf(x, wsh[j * p.c + c0 + e], acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < kMaxJoints; ++j) {
            if (j < p.joints) {
                const float v = acc[j] + p.bias[j];
                if (p.heatmaps) p.heatmaps[((size_t)n * p.joints + j) * hw + px] = v;
                if (takes(v, bv[j], bi[j])) {  // px increases monotonically per thread: strict > keeps the first maximum
                    bv[j] = v;
                    bi[j] = px;
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < kMaxJoints; ++j) {
        if (j < p.joints) {
            float v = bv[j];
            int i = bi[j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float ov = __shfl_xor(v, off);
                const int oi = __shfl_xor(i, off);
                if (better(ov, oi, v, i)) {
                    v = ov;
                    i = oi;
                }
            }
            if (lane == 0) {
                red_v[wave * kMaxJoints + j] = v;
                red_i[wave * kMaxJoints + j] = i;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < p.joints) {
        const int j = threadIdx.x;
        float v = red_v[j];
        int i = red_i[j];
        for (int w = 1; w < 4; ++w)
            if (better(red_v[w * kMaxJoints + j], red_i[w * kMaxJoints + j], v, i)) {
                v = red_v[w * kMaxJoints + j];
                i = red_i[w * kMaxJoints + j];
            }
        p.part_val[((size_t)n * p.joints + j) * p.slabs + slab] = v;
        p.part_idx[((size_t)n * p.joints + j) * p.slabs + slab] = i;
    }
}

// bf16 mode: the same head on MFMA.  D[joint][pixel] = W[joint][k] * X[k][pixel], joints padded to 32 (two
// fragments), K = c padded to a multiple of 32 (weights zero there; the x operand is forced to zero too, so stray
// bytes never meet the matrix unit).  Lane (li, g) ends up with joints 4g..4g+3 and 16+4g..16+4g+3 of pixel li:
// a running (max, first index) per slot over the wave's 16 pixel fragments, then a 16-lane butterfly over li, then
// the four waves through LDS.  wimg = [fragment][chunk][lane][8 bf16] (hrnet_mi355.cpp: load_weights).
__global__ __launch_bounds__(256) void head_mfma_kernel(const HeadArgs p) {
    __shared__ float red_v[4 * kMaxJoints];
    __shared__ int red_i[4 * kMaxJoints];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int slab = blockIdx.x, n = blockIdx.y;
    const int hw = p.h * p.w;
    const int kch = (p.c + 31) >> 5;
    const GLOBAL_AS unsigned short *__restrict__ in = (const GLOBAL_AS unsigned short *)p.in;
    const GLOBAL_AS s16x8 *__restrict__ wimg = (const GLOBAL_AS s16x8 *)p.wimg;
    float bias[8];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = f * 16 + 4 * g + r;
            bias[f * 4 + r] = j < p.joints ? p.bias[j] : 0.f;
        }
    float bv[8];
    int bi[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) bv[t] = -INFINITY, bi[t] = kNoIdx;

    const int px_per_wave = p.slab_px / 4;
    for (int it = 0; it < px_per_wave / 16; ++it) {
        const int px0 = slab * p.slab_px + wave * px_per_wave + it * 16;
        if (px0 >= hw) break;  // wave-uniform
        const int px = px0 + li;
        const bool live = px < hw;
        const int pc = live ? px : hw - 1;
        const int r = pc / p.w, c = pc - r * p.w;
        const GLOBAL_AS unsigned short *row = in + ((size_t)n * p.hpwp + r * p.wp + c) * p.c;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        for (int kc = 0; kc < kch; ++kc) {
            const int k0 = kc * 32 + g * 8;
            s16x8 x = {};
            if (k0 < p.c) x = *(const GLOBAL_AS s16x8 *)(row + k0);
#pragma unroll
            for (int f = 0; f < 2; ++f) acc[f] = mma<DT_BF16>(wimg[(f * kch + kc) * 64 + lane], x, acc[f]);
        }
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = f * 4 + q, j = f * 16 + 4 * g + q;
                const float v = acc[f][q] + bias[t];
                if (live && j < p.joints) {
                    if (p.heatmaps) p.heatmaps[((size_t)n * p.joints + j) * hw + px] = v;
                    if (takes(v, bv[t], bi[t])) bv[t] = v, bi[t] = px;  // px grows with `it`: strict > keeps the first maximum
                }
            }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        float v = bv[t];
        int i = bi[t];
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {  // over li: lanes of one k-group
            const float ov = __shfl_xor(v, off);
            const int oi = __shfl_xor(i, off);
            if (better(ov, oi, v, i)) v = ov, i = oi;
        }
        const int j = (t >> 2) * 16 + 4 * g + (t & 3);
        if (li == 0 && j < kMaxJoints) red_v[wave * kMaxJoints + j] = v, red_i[wave * kMaxJoints + j] = i;
    }
    __syncthreads();
    if (threadIdx.x < p.joints) {
        const int j = threadIdx.x;
        float v = red_v[j];
        int i = red_i[j];
        for (int w = 1; w < 4; ++w)
            if (better(red_v[w * kMaxJoints + j], red_i[w * kMaxJoints + j], v, i)) {
                v = red_v[w * kMaxJoints + j];
                i = red_i[w * kMaxJoints + j];
            }
        p.part_val[((size_t)n * p.joints + j) * p.slabs + slab] = v;
        p.part_idx[((size_t)n * p.joints + j) * p.slabs + slab] = i;
    }
}

hipError_t launch_head(int dtype, const HeadArgs &a, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    if (a.joints > kMaxJoints) return hipErrorInvalidValue;
    dim3 grid(a.slabs, a.n);
    if (dtype == DT_BF16 && a.wimg && a.slab_px % 64 == 0) {
        hipLaunchKernelGGL(head_mfma_kernel, grid, dim3(256), 0, s, a);
        return hipGetLastError();
    }
    const size_t shm = sizeof(float) * ((size_t)a.joints * a.c + 4 * kMaxJoints) + sizeof(int) * 4 * kMaxJoints;
    if (dtype == DT_BF16)
        hipLaunchKernelGGL(head_kernel<DT_BF16>, grid, dim3(256), shm, s, a);
    else
        hipLaunchKernelGGL(head_kernel<DT_F32>, grid, dim3(256), shm, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Decode (SimpleHRNet.py:297-308): merge the slab candidates (lowest flat index among equal maxima =
// np.argmax), then  y = py * 1. / h * (y2 - y1) + y1,  x = px * 1. / w * (x2 - x1) + x1  evaluated in
// float64 exactly as numpy does (box difference first, in the boxes' own dtype), stored as fp32.
__global__ void decode_kernel(const DecodeArgs p) {
#pragma clang fp contract(off)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= p.n * p.joints) return;
    const int n = t / p.joints;
    float v = -INFINITY;
    int i = kNoIdx;
    for (int s = 0; s < p.slabs; ++s) {
        const float ov = p.part_val[(size_t)t * p.slabs + s];
        const int oi = p.part_idx[(size_t)t * p.slabs + s];
        if (better(ov, oi, v, i)) {
            v = ov;
            i = oi;
        }
    }
    if (i == kNoIdx) i = 0;  // (unreachable with h*w >= 1; never form coordinates from the sentinel)
    const int py = i / p.w, px = i - py * p.w;
    double x1, y1, dx, dy;
    if (p.box_is_float) {
        const float *b = (const float *)p.boxes + 4 * (size_t)n;
        x1 = b[0], y1 = b[1];
        dx = (double)(b[2] - b[0]);  // fp32 subtraction first, like numpy float32 scalars
        dy = (double)(b[3] - b[1]);
    } else {
        const int *b = (const int *)p.boxes + 4 * (size_t)n;
        x1 = b[0], y1 = b[1];
        dx = (double)(b[2] - b[0]);
        dy = (double)(b[3] - b[1]);
    }
    float *o = p.pts + (size_t)t * 3;
    o[0] = (float)((double)py * 1. / (double)p.h * dy + y1);
    o[1] = (float)((double)px * 1. / (double)p.w * dx + x1);
    o[2] = v;
}

// Flip-TTA combine + decode (testing/Test.py:134-140, misc/utils.py:19-29, 125-175): per (crop, joint)
//   avg = (hm[j] + mirror(hm_flipped[pair(j)])) * 0.5     written back over hm
//   (max, first arg-max) of avg -> x = idx % w, y = idx / w, zeroed when max <= 0   (get_max_preds)
//   post_processing: +-0.25 px towards the higher neighbour when 1 < x < w-1 and 1 < y < h-1   (get_final_preds)
__global__ __launch_bounds__(256) void tta_decode_kernel(const TtaArgs p) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int j = blockIdx.x, n = blockIdx.y, hw = p.h * p.w;
    float *hm = p.hm + ((size_t)n * p.joints + j) * hw;
    const float *hf = p.hm_flipped + ((size_t)n * p.joints + p.pair[j]) * hw;
    float bv = -INFINITY;
    int bi = kNoIdx;
    for (int px = threadIdx.x; px < hw; px += 256) {
        const int y = px / p.w, x = px - y * p.w;
        const float v = (hm[px] + hf[y * p.w + (p.w - 1 - x)]) * 0.5f;
        hm[px] = v;
        if (takes(v, bv, bi)) bv = v, bi = px;  // px grows: strict > keeps the first maximum
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        if (better(ov, oi, bv, bi)) bv = ov, bi = oi;
    }
    if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = bv, si[threadIdx.x >> 6] = bi;
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (better(sv[w], si[w], bv, bi)) bv = sv[w], bi = si[w];
        float x = (float)(bi % p.w), y = (float)(bi / p.w);
        if (!(bv > 0.f)) x = 0.f, y = 0.f;
        if (p.post_processing) {
            const int ix = (int)x, iy = (int)y;  // integer valued
            if (1 < ix && ix < p.w - 1 && 1 < iy && iy < p.h - 1) {
                const float dx = hm[iy * p.w + ix + 1] - hm[iy * p.w + ix - 1];
                const float dy = hm[(iy + 1) * p.w + ix] - hm[(iy - 1) * p.w + ix];
                x += (dx > 0.f ? 0.25f : (dx < 0.f ? -0.25f : 0.f));
                y += (dy > 0.f ? 0.25f : (dy < 0.f ? -0.25f : 0.f));
            }
        }
        p.preds[((size_t)n * p.joints + j) * 2 + 0] = x;
        p.preds[((size_t)n * p.joints + j) * 2 + 1] = y;
        p.maxvals[(size_t)n * p.joints + j] = bv;
    }
}

hipError_t launch_tta_decode(const TtaArgs &a, hipStream_t s) {
    if (a.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(tta_decode_kernel, dim3(a.joints, a.n), dim3(256), 0, s, a);
    return hipGetLastError();
}

// Debug tap (hrn_forward_tap): the stored values of a flat padded NHWC tensor, widened exactly, as NCHW fp32.
template <int DT>
__global__ __launch_bounds__(256) void tap_kernel(const TapArgs p) {
    using T = Tr<DT>;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long per = (long)p.c * p.h * p.w;
    if (idx >= per * p.ncrops) return;
    const int n = (int)(idx / per);
    const int rem = (int)(idx - (long)n * per);
    const int ch = rem / (p.h * p.w), px = rem - ch * (p.h * p.w);
    const int r = px / p.w, c = px - r * p.w;
    const typename T::elem *in = (const typename T::elem *)p.in;
    p.dst[idx] = T::ld(in[((size_t)(p.crop0 + n * p.crop_step) * p.hpwp + (size_t)r * p.wp + c) * p.c + ch]);
}

hipError_t launch_tap(int dtype, const TapArgs &a, hipStream_t s) {
    const long total = (long)a.ncrops * a.c * a.h * a.w;
    if (total <= 0) return hipSuccess;
    dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == DT_BF16)
        hipLaunchKernelGGL(tap_kernel<DT_BF16>, grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(tap_kernel<DT_F32>, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

// Debug (hrn_debug_pad_violations): count the elements of a flat padded NHWC buffer that sit at a pad or guard position and are not
// +0 / -0: the workspace invariant every 3x3 convolution's zero padding rests on (ctx_plan.inc: new_tensor).
template <int DT>
__global__ __launch_bounds__(256) void pad_check_kernel(const PadCheckArgs p) {
    using T = Tr<DT>;
    const typename T::elem *buf = (const typename T::elem *)p.buf;
    unsigned long long bad = 0;
    for (long row = (long)blockIdx.x * 256 + threadIdx.x; row < p.rows; row += (long)gridDim.x * 256) {
        const long q = row - p.lead_rows;
        bool pad = q < 0 || q >= (long)p.nmax * p.hpwp;                 // guard rows in front of image 0 / behind the last image
        if (!pad) {
            const int rem = (int)(q % p.hpwp), r = rem / p.wp, c = rem - r * p.wp;
            pad = r >= p.h || c >= p.w;                                 // the shared pad row / pad column of an image
        }
        if (pad)
            for (int ch = 0; ch < p.c; ++ch) bad += T::ld(buf[(size_t)row * p.c + ch]) != 0.f ? 1ull : 0ull;
    }
    if (bad) atomicAdd(p.count, bad);
}

hipError_t launch_pad_check(int dtype, const PadCheckArgs &a, hipStream_t s) {
    if (a.rows <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((a.rows + 255) / 256 < 4096 ? (a.rows + 255) / 256 : 4096);
    if (dtype == DT_BF16)
        hipLaunchKernelGGL(pad_check_kernel<DT_BF16>, dim3(blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(pad_check_kernel<DT_F32>, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_decode(const DecodeArgs &a, hipStream_t s) {
    const int total = a.n * a.joints;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(decode_kernel, dim3((total + 127) / 128), dim3(128), 0, s, a);
    return hipGetLastError();
}

}  // namespace hrn
