// The stem as ONE kernel (round 4): conv1 (3 -> 64, 3x3 stride 2, BN, ReLU; models_/hrnet.py:79-80,158-160) computed into LDS and
// conv2 (64 -> 64, 3x3 stride 2, BN, ReLU; :81-83,161-163) read from there -- bf16 mode, written for gfx950.
//
// Separately the two convolutions write and read back the largest tensor of the pass (64 channels at half resolution: 0.9 GB
// at 256 crops of 384x288) and took 0.43 + 0.38 ms of a 23.6-ms pass, both HBM-bound.  Here:
//   * tile = ONE output row of conv2 of one image (R = 1 of the stride-2 slab kernel, conv_s2.hip): its input footprint is three
//     rows of conv1's output, which need seven rows of the crop.
//   * the crop rows come in as coalesced float4 loads, issued a whole tile before they are used (registers), are rounded to bf16 exactly as
//     stem_mfma_kernel rounds them and kept in LDS as [row][colour][column] (the "patch").
//   * phase A: conv1 on MFMA (K = 27 padded to one 32-wide chunk, the same weight image, bias-initialised accumulators, ReLU,
//     bf16 rounding as stem_mfma_kernel) for the 3 x 2 x Wop slots of the stride-2 slab -- DIRECTLY in the slab's layout
//     (de-interleaved by column parity, 32-byte sub-slots in four regions, bank pad per row pair: conv_s2.hip / kernels.h), with
//     exact zeros where conv2's padding is.  A lane gathers its eight k-values of one conv1 pixel from the patch with
//     ds_read_u16 (pairs packed with one v_lshl_or each).
//   * phase B: conv2 exactly as s2_run<64, 2, 2> does it: this wave's 32 x 576 weight matrix resident in registers (144 VGPRs),
//     no barrier in the K loop, one ds_read_b128 per two MFMAs, bias / ReLU / pad-column epilogue.
//   * two slab buffers and two patch buffers, ONE barrier per tile.
// Same arithmetic as the two kernels it replaces (K orders, accumulator initialisation, roundings) -> results are BIT-IDENTICAL
// to stem_mfma_kernel + the generic / slab kernel; tests/test_stem_fused.py checks that on the whole net.  conv1's output never
// exists in HBM (its debug tap "stem" makes the handle take the two-kernel path for that call).
#include "kernels.h"

namespace hrn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#define GLOBAL_AS __attribute__((address_space(1)))

namespace {

constexpr int SF_REGION = kStemFuseRegionBytes;        // one 16-channel region of a slab buffer: 32-byte sub-slots
constexpr int SF_SLAB = 4 * SF_REGION;                 // one slab buffer (64 channels)
constexpr int SF_BIAS = 2 * SF_SLAB;                   // conv2's 64 biases (fp32)
constexpr int SF_BIAS1 = SF_BIAS + 256;                // conv1's 64 biases (fp32)
constexpr int SF_W1 = SF_BIAS1 + 256;                  // conv1's weight image (4 fragments x 64 lanes x 16 bytes)
constexpr int SF_PATCH = SF_W1 + 4096;                 // two patch buffers
constexpr int SF_LDS = SF_PATCH + 2 * kStemFusePatchBytes;
static_assert(SF_LDS <= 160 * 1024, "LDS budget");

__device__ __forceinline__ unsigned short sf_bf16(float f) {  // round to nearest even, as kernels.hip: f32_to_bf16
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__device__ __forceinline__ void stemf_run(const GLOBAL_AS S2Problem *pp, const StemArgs st, const int ntile, const int tile0, char *smem) {
    constexpr int NF = 2, MW = 2, ROWB = 32, NCH = 18, CPP = 32, NT = 512;
    // everything the loop needs as scalars, once (no kernel-argument / descriptor load may be in flight while counted lgkmcnt
    // waits run: SMEM returns out of order)
    const int Ho = pp->ho, Wo = pp->wo, Wop = pp->wop, out_hpwp = pp->out_hpwp, nparts = pp->nparts;
    const unsigned magic_wop = pp->magic_wop;
    const int shift_wop = pp->shift_wop;
    const int H = st.H, W = st.W, flip = st.flip;
    const int h1 = st.out_h, wd1 = st.out_w;                       // conv1's output grid
    const GLOBAL_AS float *const images = (const GLOBAL_AS float *)st.images;
    const int PW = W + 8;                                         // patch row pitch (elements): column x at index x + 4
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int wpart = __builtin_amdgcn_readfirstlane((int)pp->wave_part[wave]);
    const int wf0 = __builtin_amdgcn_readfirstlane((int)pp->wave_f0[wave]), wfs = __builtin_amdgcn_readfirstlane((int)pp->wave_fs[wave]);
    const bool active = wpart < nparts;
    const int part = active ? wpart : 0;
    const int spv = 2 * Wop, PP = s2_pair_pitch(Wop);             // slots per virtual row / per row pair (with the bank pad)
    const int slots = PP + spv;                                   // three virtual rows
    const int frags1 = (slots + 15) >> 4;                         // conv1 pixel fragments per tile
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    float *const bias_lds = (float *)(smem + SF_BIAS);

    // ---- the block's constants: conv2 weights of this wave's part (registers), conv1's weight image and biases (registers),
    //      conv2's biases (LDS), zeros in the patch buffers (their edge columns are never written again)
    s16x8 wf[NCH][NF];
    {
        const GLOBAL_AS char *wsrc = (const GLOBAL_AS char *)pp->part[part].w + lane * 16;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int j = 0; j < NF; ++j) wf[c][j] = *(const GLOBAL_AS s16x8 *)(wsrc + (c * NF + j) * 1024);
        if (tid < nparts * CPP) {
            const int pt = tid / CPP, ch = tid - pt * CPP;
            bias_lds[pt * CPP + ch] = ((const GLOBAL_AS float *)pp->part[pt].bias)[pp->part[pt].ch0 + ch];
        }
    }
    if (tid < 256) ((u32x4 *)(smem + SF_W1))[tid] = *(const GLOBAL_AS u32x4 *)((const GLOBAL_AS char *)st.wp + tid * 16);
    const s16x8 *const w1 = (const s16x8 *)(smem + SF_W1) + lane;        // fragment jj at w1[64 jj]
    if (tid < 64) ((float *)(smem + SF_BIAS1))[tid] = ((const GLOBAL_AS float *)st.bias)[tid];
    const f32x4 *const b1 = (const f32x4 *)(smem + SF_BIAS1) + g * 4;   // accumulator row 4 jj + r of k-group g = channel 16 g + 4 jj + r
    for (int i = tid; i < 2 * kStemFusePatchBytes / 4; i += NT) ((unsigned *)(smem + SF_PATCH))[i] = 0u;
    const int cout = pp->part[part].cout, ch0 = pp->part[part].ch0, relu = pp->part[part].relu;
    GLOBAL_AS unsigned short *const out = (GLOBAL_AS unsigned short *)pp->part[part].out;
    const float lo = relu ? 0.f : -INFINITY;

    // this lane's eight k-values of a conv1 pixel: k = 8 g + e -> (colour, kh, kw); k >= 27: the zero padding of the chunk.
    // Patch element of (slab row v, plane, j): row 2 v + kh, colour, column index 4 j + 2 plane + kw + 1 (see the header)
    int koff[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = g * 8 + e;
        const int ci = k / 9, kh = (k % 9) / 3, kw = k % 3;
        koff[e] = k < 27 ? ((kh * 3 + ci) * PW + kw + 1) * 2 : -1;
    }
    // conv2: LDS byte offset of k-group g of chunk-within-tap `sub` relative to a pixel's slot in region 0
    int xoff[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        const int ci = 32 * sub + 8 * g;
        xoff[sub] = (ci / 16) * SF_REGION + (ci % 16) * 2;
    }
    auto chunk_off = [&](int c) -> unsigned {   // c is a compile-time constant at every call
        const int tap = c / 2, dh = tap / 3, dw = tap - 3 * dh;
        return (unsigned)(((dh == 2 ? PP : dh * spv) + (dw & 1) * Wop + (dw >> 1)) * ROWB) + (unsigned)xoff[c % 2];
    };

    // ---- the crop rows of a tile: (row r = 0..6, colour) lines of W floats = 21 W / 4 float4s, up to three per thread
    const int w4 = W >> 2, nvec = 21 * w4;
    // The loads are asm: the compiler's vmcnt bookkeeping is not path-sensitive and would wait for them (and for the epilogue's
    // stores) at the top of every loop; here ONE wait per tile, just before the values are used, a whole tile after the issue.
    // An asm load's result register must not be copied before the wait (the compiler believes the asm has completed): the
    // loads are therefore UNCONDITIONAL straight-line code -- clamped addresses, rows outside the crop zeroed when the values
    // are stored -- so that no control-flow merge makes the compiler move them.
    f32x4 pre[3];
    auto fetch_patch = [&](int t) {
        const int n = t / Ho, ho = t - n * Ho;
        const int y0 = 4 * ho - 3;
        const GLOBAL_AS float *img = images + (size_t)n * 3 * H * W;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            int idx = tid + q * NT;
            idx = idx < nvec ? idx : nvec - 1;
            const int line = idx / w4, x4 = idx - line * w4;
            const int r = line / 3, ci = line - r * 3;
            int y = y0 + r;
            y = y < 0 ? 0 : (y >= H ? H - 1 : y);
            const GLOBAL_AS float *src = img + ((size_t)ci * H + y) * W + 4 * x4;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pre[q]) : "v"(src));
        }
    };
    auto fetch_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" : "+v"(pre[0]), "+v"(pre[1]), "+v"(pre[2])); };
    auto store_patch = [&](int b, int t) {
        char *pb = smem + SF_PATCH + b * kStemFusePatchBytes;
        const int ho = t - (t / Ho) * Ho;
        const int y0 = 4 * ho - 3;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int idx = tid + q * NT;
            if (idx < nvec) {
                const int line = idx / w4, x4 = idx - line * w4;
                const int y = y0 + line / 3;
                const bool in = y >= 0 && y < H;
                const unsigned short a0 = in ? sf_bf16(pre[q][0]) : 0, a1 = in ? sf_bf16(pre[q][1]) : 0, a2 = in ? sf_bf16(pre[q][2]) : 0,
                                     a3 = in ? sf_bf16(pre[q][3]) : 0;
                // column x at index x + 4; mirrored crops (flip-TTA): column x holds pixel W - 1 - x
                const int col = flip ? W - 4 - 4 * x4 : 4 * x4;
                const unsigned lo2 = flip ? ((unsigned)a3 | ((unsigned)a2 << 16)) : ((unsigned)a0 | ((unsigned)a1 << 16));
                const unsigned hi2 = flip ? ((unsigned)a1 | ((unsigned)a0 << 16)) : ((unsigned)a2 | ((unsigned)a3 << 16));
                *(u32x2 *)(pb + ((size_t)line * PW + col + 4) * 2) = u32x2{lo2, hi2};
            }
        }
    };
    const int tlast = tile0 + ntile - 1;
    fetch_patch(tile0);
    __syncthreads();          // the zeroed patch buffers, conv2's biases
    fetch_wait();
    store_patch(0, tile0);
    fetch_patch(tile0 + 1 < tlast ? tile0 + 1 : tlast);   // (always issued: see above; past the run's end the last tile again, unused)
    __syncthreads();

    for (int k = 0; k < ntile; ++k) {
        const int t = tile0 + k, b = k & 1;
        const int n = t / Ho, ho = t - n * Ho;
        // ---- phase A: conv1 for the slab's slots, fragments wave, wave + 8, ...
        {
            const unsigned pbase = lds0 + SF_PATCH + b * kStemFusePatchBytes;
            char *const sl = smem + b * SF_SLAB;
            for (int f = wave; f < frags1; f += 8) {
                const int s = f * 16 + li;
                // slot -> (virtual row v, plane, j); pad slots and slots past the end compute garbage that is replaced by zeros
                const int pr = s >= PP ? 1 : 0, o = s - pr * PP;
                const int second = (pr == 0 && o >= spv) ? 1 : 0;
                int rem = o - second * spv;
                const bool in_row = rem < spv && s < slots;
                if (rem >= spv) rem = spv - 1;
                const int v = 2 * pr + second;
                const int plane = rem >= Wop ? 1 : 0, j = rem - plane * Wop;
                const int r1 = 2 * ho - 1 + v, c1 = 2 * j - 1 + plane;
                const bool ok = in_row && r1 >= 0 && r1 < h1 && c1 >= 0 && c1 < wd1;
                const unsigned pa = pbase + (unsigned)((2 * v * 3 * PW + 4 * j + 2 * plane) * 2);
                // (no ds_read_u16_d16 / _d16_hi pairs: with SRAM ECC on -- every MI300 / MI355 -- a d16 load rewrites the WHOLE register)
                unsigned xe[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    // k >= 27 (the last five values of k-group 3): element 0 of the patch, a zero that is never overwritten
                    const unsigned ad = koff[e] >= 0 ? pa + (unsigned)koff[e] : pbase;
                    asm volatile("ds_read_u16 %0, %1" : "=v"(xe[e]) : "v"(ad));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xe[0]), "+v"(xe[1]), "+v"(xe[2]), "+v"(xe[3]), "+v"(xe[4]), "+v"(xe[5]), "+v"(xe[6]), "+v"(xe[7]));
                unsigned xw[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) xw[e] = xe[2 * e] | (xe[2 * e + 1] << 16);
                const s16x8 xf = __builtin_bit_cast(s16x8, (u32x4{xw[0], xw[1], xw[2], xw[3]}));
                f32x4 acc[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w1[64 * jj]), __builtin_bit_cast(bf16x8, xf), b1[jj], 0, 0, 0);
                // ReLU, bf16, zeros where conv2's padding is; the lane owns channels 16 g .. 16 g + 15 of its pixel = its 32-byte
                // sub-slot in region g
                unsigned pk[8];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float a0 = ok ? fmaxf(acc[jj][2 * h], 0.f) : 0.f, a1 = ok ? fmaxf(acc[jj][2 * h + 1], 0.f) : 0.f;
                        pk[2 * jj + h] = (unsigned)sf_bf16(a0) | ((unsigned)sf_bf16(a1) << 16);
                    }
                if (s < slots) {
                    char *d = sl + g * SF_REGION + s * ROWB;
                    *(u32x4 *)d = u32x4{pk[0], pk[1], pk[2], pk[3]};
                    *(u32x4 *)(d + 16) = u32x4{pk[4], pk[5], pk[6], pk[7]};
                }
            }
        }
        // the crop rows of tile k + 1 (issued a whole tile ago) -> the other patch buffer; then those of tile k + 2 go out: in
        // flight under phase B and the next phase A
        fetch_wait();
        if (k + 1 < ntile) store_patch(b ^ 1, t + 1);
        fetch_patch(t + 2 < tlast ? t + 2 : tlast);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();               // the slab of this tile and the patch of the next one are complete

        // ---- phase B: conv2 over the slab (s2_run<64, 2, 2> with one output row per tile)
        const int npx = Wop;
        const int mf = (npx + 15) >> 4;
        const long q0 = (long)n * out_hpwp + (long)ho * Wop;
        if (active) {
            for (int f0 = wf0; f0 < mf; f0 += MW * wfs) {
                int nfr = 0;
                int tp[MW], wo[MW];
                unsigned xa[MW];
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    if (f0 + i * wfs < mf) nfr = i + 1;
                    int tt = (f0 + i * wfs) * 16 + li;
                    tp[i] = tt;
                    if (tt >= npx) tt = 0;
                    const int rr = (int)(((unsigned long long)(unsigned)tt * magic_wop) >> shift_wop);   // (0: one row per tile)
                    wo[i] = tt - rr * Wop;
                    xa[i] = lds0 + b * SF_SLAB + (rr * PP + wo[i]) * ROWB;
                }
                f32x4 acc[MW][NF];
#pragma unroll
                for (int i = 0; i < MW; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                s16x8 xf[2][MW];
#define SF_READ(SET, C)                                                                                         \
    {                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < MW; ++i)                                                          \
            asm volatile("ds_read_b128 %0, %1" : "=v"(xf[SET][i]) : "v"(xa[i] + chunk_off(C)));                 \
    }
                SF_READ(0, 0)
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int cur = c & 1;
                    if (c + 1 < NCH) {
                        SF_READ(cur ^ 1, c + 1)
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(MW) : "memory");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NF; ++j)
#pragma unroll
                        for (int i = 0; i < MW; ++i)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[c][j]),
                                                                                __builtin_bit_cast(bf16x8, xf[cur][i]), acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef SF_READ
                // ---- epilogue (as s2_run): + bias, ReLU, zero at the pad column; a lane owns 8 contiguous channels of one pixel
                const float *bl = bias_lds + part * CPP + g * 4 * NF;
                f32x4 bs[NF];
#pragma unroll
                for (int j = 0; j < NF; ++j) bs[j] = *(const f32x4 *)(bl + 4 * j);
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    if (i >= nfr) break;
                    if (tp[i] < npx) {
                        const float hi = wo[i] < Wo ? INFINITY : 0.f;
                        const float lo_i = wo[i] < Wo ? lo : 0.f;
                        unsigned pk2[2 * NF];
#pragma unroll
                        for (int j = 0; j < NF; ++j)
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                float a0 = acc[i][j][2 * h] + bs[j][2 * h], a1 = acc[i][j][2 * h + 1] + bs[j][2 * h + 1];
                                asm("v_med3_f32 %0, %1, %2, %3" : "=v"(a0) : "v"(a0), "v"(lo_i), "v"(hi));
                                asm("v_med3_f32 %0, %1, %2, %3" : "=v"(a1) : "v"(a1), "v"(lo_i), "v"(hi));
                                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk2[2 * j + h]) : "v"(a0), "v"(a1));
                            }
                        GLOBAL_AS unsigned short *o = out + (size_t)(q0 + tp[i]) * cout + ch0 + g * 4 * NF;
                        *(GLOBAL_AS u32x4 *)o = u32x4{pk2[0], pk2[1], pk2[2], pk2[3]};
                    }
                }
            }
        }
    }
}

}  // namespace

__global__ __launch_bounds__(512) void stem_fused_kernel(const S2Problem *__restrict__ probs, const int2 *__restrict__ map, const StemArgs stem) {
    extern __shared__ __attribute__((aligned(1024))) char smem_sf[];
    const int2 e = map[blockIdx.x];
    const int prob = __builtin_amdgcn_readfirstlane(e.x & 0xff), ntile = __builtin_amdgcn_readfirstlane(e.x >> 8);
    const int tile0 = __builtin_amdgcn_readfirstlane(e.y);
    stemf_run((const GLOBAL_AS S2Problem *)(probs + prob), stem, ntile, tile0, smem_sf);
}

// the fused stem handles this geometry (conv2's output width + pad column, the crop width): its slab regions and patch buffers hold it
int stem_fused_fits(int wop, int w_in) {
    return w_in % 4 == 0 && s2_pair_pitch(wop) + 2 * wop <= kStemFuseRegionBytes / 32 && 21 * (w_in + 8) * 2 <= kStemFusePatchBytes;
}

hipError_t launch_stem_fused(const S2Problem *probs_dev, const void *map_dev, int nblocks, const StemArgs &stem, hipStream_t s) {
    if (nblocks <= 0) return hipSuccess;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)stem_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SF_LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(stem_fused_kernel, dim3(nblocks), dim3(512), SF_LDS, s, probs_dev, (const int2 *)map_dev, stem);
    return hipGetLastError();
}

}  // namespace hrn
