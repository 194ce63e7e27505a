// The stem as ONE kernel (round 4): conv1 (3 -> 64, 3x3 stride 2, BN, ReLU; models_/hrnet.py:79-80,158-160) computed into LDS and
// conv2 (64 -> 64, 3x3 stride 2, BN, ReLU; :81-83,161-163) read from there -- bf16 mode, written for gfx950.
//
// Separately the two convolutions write and read back the largest tensor of the pass (64 channels at half resolution: 0.9 GB
// at 256 crops of 384x288) and took 0.43 + 0.38 ms of a 23.6-ms pass, both HBM-bound; this kernel takes 0.34 ms.
//   * tile = ONE output row of conv2 of one image (R = 1 of the stride-2 slab kernel, conv_s2.hip): its input footprint is three
//     rows of conv1's output, which need seven rows of the crop.
//   * the crop rows come in as coalesced float4 loads issued a whole tile before they are used (registers; asm, one explicit
//     vmcnt wait per tile), are rounded to bf16 exactly as stem_mfma_kernel rounds them and kept in LDS as
//     [row][colour][column] (the "patch").
//   * phase A: conv1 on MFMA (K = 27 padded to one 32-wide chunk, the same weight image, bias-initialised accumulators, ReLU,
//     bf16 rounding as stem_mfma_kernel) for the 3 x 2 x Wop slots of the stride-2 slab -- DIRECTLY in the slab's layout
//     (de-interleaved by column parity, 32-byte sub-slots in four regions, bank pad per row pair: conv_s2.hip / kernels.h), with
//     exact zeros where conv2's padding is.  A lane gathers its eight k-values of one conv1 pixel from the patch with
//     ds_read_u16 (pairs packed with one v_lshl_or each), one fragment ahead of the MFMAs.
//   * phase B: conv2 in the arithmetic of s2_run<64, 2, *>: this wave's 32 x 576 weight matrix resident in registers
//     (144 VGPRs), no barrier in the K loop, one ds_read_b128 per two MFMAs (three chunks ahead), bias / ReLU / pad-column
//     epilogue; one pixel fragment at a time, only the fragments the row has.
//   * two slab buffers and two patch buffers, ONE barrier per tile.
// Same arithmetic as the two kernels it replaces (K orders, accumulator initialisation, roundings) -> results are BIT-IDENTICAL
// to stem_mfma_kernel + the generic / slab kernel for finite inputs; tests/test_stem_fused.py checks that on the whole net.
// conv1's output never exists in HBM (its debug tap "stem" makes the handle take the two-kernel path for that call).
//
// Where the time goes (per-phase shader-clock stamps of a round-4 debug build, 256 crops of 384x288, shader clocks per tile of a
// ~7200-clock tile): phase A ~2100-2700, patch store + next loads ~1300, phase B ~2000-2600, barrier wait ~1100.  No unit is
// saturated (LDS pipe ~55 % busy, VALU ~35 %, MFMA ~25 %): with 144 weight VGPRs per wave the CU holds two waves per SIMD, and
// every phase is a dependent chain (gather -> MFMA -> pack -> LDS write; LDS read -> MFMA) that two waves cannot cover.
// profiles/EXPERIMENTS.md (round 4) has the steps that got it from 643 us to here and what each was worth.
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

__device__ __forceinline__ void stemf_run(const GLOBAL_AS S2Problem *pp, const StemArgs st, const int ntile, const int tile0, char *smem) {
    constexpr int NF = 2, ROWB = 32, NCH = 18, CPP = 32, NT = 512;
    // everything the loop needs as scalars, once (no kernel-argument / descriptor load may be in flight while counted lgkmcnt
    // waits run: SMEM returns out of order)
    const int Ho = pp->ho, Wo = pp->wo, Wop = pp->wop, out_hpwp = pp->out_hpwp, nparts = pp->nparts;
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
    // This wave's conv1 fragments of a tile: slab slots 16 (awave + 8 i) + li, i = 0..3 (at most 30 fragments: kernels.h).  Slot ->
    // (virtual row v, column-parity plane, j): where the window of its conv1 pixel starts in a patch (row 2 v, column index
    // 4 j + 2 plane: see koff below), whether it is a pixel of conv1's grid at all, whether the slot exists -- the same for
    // every tile: patch offset | v << 16 | (column inside the grid) << 18 | (slot exists) << 19
    // Which wave takes fragments a, a + 8, ...: the waves that have ONE pixel fragment of conv2 (phase B) come first -- the
    // conv1 fragments past a multiple of eight go to them, not to the wave of each part that has two (28 + 10 fragments over
    // 8 waves: 5 + 5 + 5 + 5 + 5 + 5 + 4 + 4 instead of 6 + 5 + 5 + 5 + 4 + 4 + 5 + 4 -- the barrier waits for the slowest)
    constexpr int NFA = 4;
    int awave = 0;
    {
        const int mfb = (Wop + 15) >> 4;
        int light_before = 0, heavy_before = 0, light_total = 0;
        bool me_heavy = false;
        for (int u = 0; u < 8; ++u) {
            const int up = __builtin_amdgcn_readfirstlane((int)pp->wave_part[u]);
            const int uf0 = __builtin_amdgcn_readfirstlane((int)pp->wave_f0[u]), ufs = __builtin_amdgcn_readfirstlane((int)pp->wave_fs[u]);
            const bool heavy = up < nparts && uf0 + ufs < mfb;
            if (u == wave) me_heavy = heavy;
            if (u < wave) (heavy ? heavy_before : light_before) += 1;
            if (!heavy) light_total += 1;
        }
        awave = me_heavy ? light_total + heavy_before : light_before;
    }
    const bool has_last = (awave + 8 * (NFA - 1)) * 16 < slots;      // (wave-uniform) the fourth fragment exists
    unsigned fa[NFA];
#pragma unroll
    for (int i = 0; i < NFA; ++i) {
        const int s = (awave + 8 * i) * 16 + li;
        const int pr = s >= PP ? 1 : 0, o = s - pr * PP;
        const int second = (pr == 0 && o >= spv) ? 1 : 0;
        int rem = o - second * spv;
        const bool exists = s < slots;
        const bool in_row = rem < spv && exists;          // (not the bank pad between the row pairs, not past the end)
        if (rem >= spv) rem = spv - 1;
        const int v = exists ? 2 * pr + second : 0;
        const int plane = rem >= Wop ? 1 : 0, j = exists ? rem - plane * Wop : 0;
        const int c1 = 2 * j - 1 + plane;
        const bool colok = in_row && c1 >= 0 && c1 < wd1;
        fa[i] = (unsigned)((2 * v * 3 * PW + 4 * j + 2 * plane) * 2) | ((unsigned)v << 16) | (colok ? 1u << 18 : 0u) | (exists ? 1u << 19 : 0u);
    }
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
    auto tap_off = [&](int c) -> unsigned {     // c is a compile-time constant at every call; the value is wave-uniform
        const int tap = c / 2, dh = tap / 3, dw = tap - 3 * dh;
        return (unsigned)(((dh == 2 ? PP : dh * spv) + (dw & 1) * Wop + (dw >> 1)) * ROWB);
    };

    // ---- the crop rows of a tile: (row r = 0..6, colour) lines of W floats = 21 W / 4 float4s, up to three per thread.  Which
    //      float4 a thread moves does not depend on the tile: its place in the crop (without the row term) and in the patch,
    //      computed once (the integer divisions were a fifth of the kernel's time)
    const int w4 = W >> 2, nvec = 21 * w4;
    int f_src[3];        // byte offset within the crop of the thread's float4, less its row term: ((ci H) W + 4 x4) * 4
    unsigned f_dst[3];   // byte offset in a patch buffer | r << 20 | (nothing to store) << 31
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        int idx = tid + q * NT;
        const bool live = idx < nvec;
        idx = live ? idx : nvec - 1;          // (the load is unconditional: see below)
        const int line = idx / w4, x4 = idx - line * w4;
        const int r = line / 3, ci = line - r * 3;
        f_src[q] = (ci * H * W + 4 * x4) * 4;
        // column x at index x + 4; mirrored crops (flip-TTA): column x holds pixel W - 1 - x
        const int col = flip ? W - 4 - 4 * x4 : 4 * x4;
        f_dst[q] = (unsigned)((line * PW + col + 4) * 2) | ((unsigned)r << 20) | (live ? 0u : 0x80000000u);
    }
    // The loads are asm: the compiler's vmcnt bookkeeping is not path-sensitive and would wait for them (and for the epilogue's
    // stores) at the top of every loop; here ONE wait per tile, just before the values are used, a whole tile after the issue.
    // An asm load's result register must not be copied before the wait (the compiler believes the asm has completed): the
    // loads are therefore UNCONDITIONAL straight-line code -- clamped addresses, rows outside the crop zeroed when the values
    // are stored -- so that no control-flow merge makes the compiler move them.
    f32x4 pre[3];
    auto fetch_patch = [&](int t) {
        const int n = t / Ho, ho = t - n * Ho;
        const int y0 = 4 * ho - 3;
        const GLOBAL_AS char *img = (const GLOBAL_AS char *)(images + (size_t)n * 3 * H * W);   // (wave-uniform: an SGPR pair)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            // rows above / below the crop: clamped to a valid row (zeroed at the store)
            unsigned fd = f_dst[q];
            asm volatile("" : "+v"(fd));     // (opaque: or the compiler keeps every field of it in a register of its own -> spills)
            int y = y0 + (int)((fd >> 20) & 7u);
            y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
            const int off = f_src[q] + y * (W * 4);
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(pre[q]) : "v"(off), "s"(img));
        }
    };
    auto fetch_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" : "+v"(pre[0]), "+v"(pre[1]), "+v"(pre[2])); };
    auto store_patch = [&](int b, int t) {
        char *pb = smem + SF_PATCH + b * kStemFusePatchBytes;
        const int ho = t - (t / Ho) * Ho;
        const int y0 = 4 * ho - 3;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            unsigned fd = f_dst[q];
            asm volatile("" : "+v"(fd));
            if ((int)fd >= 0) {
                const int y = y0 + (int)((fd >> 20) & 7u);
                const bool in = y >= 0 && y < H;
                // rounded to bf16 as stem_mfma_kernel rounds its inputs (nearest even); mirrored: the four pixels in reverse
                const float x0 = flip ? pre[q][3] : pre[q][0], x1 = flip ? pre[q][2] : pre[q][1];
                const float x2 = flip ? pre[q][1] : pre[q][2], x3 = flip ? pre[q][0] : pre[q][3];
                unsigned lo2, hi2;
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo2) : "v"(x0), "v"(x1));
                asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi2) : "v"(x2), "v"(x3));
                *(u32x2 *)(pb + (fd & 0xfffffu)) = u32x2{in ? lo2 : 0u, in ? hi2 : 0u};
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
            // conv1's weights and biases: from LDS once per tile (per fragment they were half of phase A's LDS traffic, and LDS
            // bandwidth is what bounds phase A); not kept across phase B: its weights need the registers
            s16x8 w1r[4];
            f32x4 b1r[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) w1r[jj] = w1[64 * jj], b1r[jj] = b1[jj];
            // The gather of fragment i + 1 is in flight while fragment i is multiplied, clamped and stored: straight-line code
            // (four fragments, the ones a wave does not have are computed on slot 0's window and not stored) so that the asm
            // loads' registers are never copied while a load is outstanding.
            // (no ds_read_u16_d16 / _d16_hi pairs: with SRAM ECC on -- every MI300 / MI355 -- a d16 load rewrites the WHOLE register)
            unsigned xe[8];
            auto gather = [&](int i) {
                unsigned fd = fa[i];
                asm volatile("" : "+v"(fd));
                const unsigned pa = pbase + (fd & 0xffffu);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    // k >= 27 (the last five values of k-group 3): element 0 of the patch, a zero that is never overwritten
                    const unsigned ad = koff[e] >= 0 ? pa + (unsigned)koff[e] : pbase;
                    asm volatile("ds_read_u16 %0, %1" : "=v"(xe[e]) : "v"(ad));
                }
            };
            gather(0);
#pragma unroll
            for (int i = 0; i < NFA; ++i) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xe[0]), "+v"(xe[1]), "+v"(xe[2]), "+v"(xe[3]), "+v"(xe[4]), "+v"(xe[5]), "+v"(xe[6]), "+v"(xe[7]));
                unsigned xw[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) xw[e] = xe[2 * e] | (xe[2 * e + 1] << 16);
                asm volatile("" : "+v"(xw[0]), "+v"(xw[1]), "+v"(xw[2]), "+v"(xw[3]));   // (packed BEFORE xe is reused)
                if (i + 1 < NFA) gather(i + 1);     // (unconditional: no control flow around an outstanding asm load)
                if (i == NFA - 1 && !has_last) break;
                unsigned fd = fa[i];
                asm volatile("" : "+v"(fd));
                const int v = (int)((fd >> 16) & 3u);
                const bool ok = (fd & (1u << 18)) && (unsigned)(2 * ho - 1 + v) < (unsigned)h1;   // a pixel of conv1's grid (else: conv2's padding)
                const s16x8 xf = __builtin_bit_cast(s16x8, (u32x4{xw[0], xw[1], xw[2], xw[3]}));
                f32x4 acc[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w1r[jj]), __builtin_bit_cast(bf16x8, xf), b1r[jj], 0, 0, 0);
                // ReLU (clamp to [0, inf]) or, where conv2's padding is, exact zeros (clamp to [0, 0]); bf16 (nearest even, as
                // stem_mfma_kernel); the lane owns channels 16 g .. 16 g + 15 of its pixel = its 32-byte sub-slot in region g
                const float hi = ok ? INFINITY : 0.f;
                unsigned pk[8];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        // (the builtin, not asm: the compiler has to see this first read of the MFMA results to space it)
                        const float a0 = __builtin_amdgcn_fmed3f(acc[jj][2 * h], 0.f, hi), a1 = __builtin_amdgcn_fmed3f(acc[jj][2 * h + 1], 0.f, hi);
                        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[2 * jj + h]) : "v"(a0), "v"(a1));
                    }
                if (fd & (1u << 19)) {
                    char *d = sl + g * SF_REGION + ((awave + 8 * i) * 16 + li) * ROWB;
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
            // One pixel fragment (16 pixels x this wave's 32 couts) at a time, only the fragments the row has (5 at 288 columns:
            // one wave of each part takes two): the B operand comes from LDS once per TWO MFMAs, which at four busy SIMDs is all
            // the LDS delivers (128 B / clock) -- a fragment nobody needs would cost what a real one does.  Reads run three
            // chunks ahead of the MFMAs (LDS latency under this load is ~200 clocks = six MFMAs).
            for (int f = wf0; f < mf; f += wfs) {
                const int tp = f * 16 + li;
                const int wo = tp < npx ? tp : 0;
                // the two per-lane bases (k-group g of the first / second half of a tap's 64 channels); opaque, or the compiler
                // keeps xoff + tap_off(c) for all 18 chunks in registers of their own across the whole kernel (-> spills)
                unsigned xs[2] = {lds0 + b * SF_SLAB + wo * ROWB + (unsigned)xoff[0], lds0 + b * SF_SLAB + wo * ROWB + (unsigned)xoff[1]};
                asm volatile("" : "+v"(xs[0]), "+v"(xs[1]));
                f32x4 acc[NF];
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                s16x8 xf[4];
#define SF_READ(C) asm volatile("ds_read_b128 %0, %1" : "=v"(xf[(C) & 3]) : "v"(xs[(C) & 1] + tap_off(C)));
                SF_READ(0)
                SF_READ(1)
                SF_READ(2)
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (c + 3 < NCH) {
                        SF_READ(c + 3)
                        asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                    } else if (c + 3 == NCH) {
                        asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                    } else if (c + 2 == NCH) {
                        asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NF; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[c][j]), __builtin_bit_cast(bf16x8, xf[c & 3]), acc[j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef SF_READ
                // ---- epilogue (as s2_run): + bias, ReLU, zero at the pad column; a lane owns 8 contiguous channels of one pixel
                const float *bl = bias_lds + part * CPP + g * 4 * NF;
                f32x4 bs[NF];
#pragma unroll
                for (int j = 0; j < NF; ++j) bs[j] = *(const f32x4 *)(bl + 4 * j);
                if (tp < npx) {
                    const float hi = wo < Wo ? INFINITY : 0.f;
                    const float lo_i = wo < Wo ? lo : 0.f;
                    unsigned pk2[2 * NF];
#pragma unroll
                    for (int j = 0; j < NF; ++j)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            float a0 = acc[j][2 * h] + bs[j][2 * h], a1 = acc[j][2 * h + 1] + bs[j][2 * h + 1];
                            asm("v_med3_f32 %0, %1, %2, %3" : "=v"(a0) : "v"(a0), "v"(lo_i), "v"(hi));
                            asm("v_med3_f32 %0, %1, %2, %3" : "=v"(a1) : "v"(a1), "v"(lo_i), "v"(hi));
                            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk2[2 * j + h]) : "v"(a0), "v"(a1));
                        }
                    GLOBAL_AS unsigned short *o = out + (size_t)(q0 + tp) * cout + ch0 + g * 4 * NF;
                    *(GLOBAL_AS u32x4 *)o = u32x4{pk2[0], pk2[1], pk2[2], pk2[3]};
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
    static std::atomic<unsigned long long> lds_set{0};   // per device: kernels.h set_dynamic_lds
    {
        const hipError_t e = set_dynamic_lds((const void *)stem_fused_kernel, SF_LDS, lds_set);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(stem_fused_kernel, dim3(nblocks), dim3(512), SF_LDS, s, probs_dev, (const int2 *)map_dev, stem);
    return hipGetLastError();
}

}  // namespace hrn

