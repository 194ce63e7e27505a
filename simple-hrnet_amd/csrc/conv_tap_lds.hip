// Tap-list LDS-staged 3x3 convolution for gfx950 (bf16 MFMA): every 3x3 conv of the graph that is NOT a
// BasicBlock conv -- stride-2 convs (stem conv2, transitions, fuse-down chains) and stride-1 convs whose channel
// count is not a multiple of 48 (layer1's 64->64, transition1's 256->48).
//
// The GEMM K dimension is cut into "slices"; a slice is (source view, channel range, tap list):
//   * stride 1: view = the input tensor itself (same geometry as the output), taps = all nine, row shift
//     (kh-1)*Wp + (kw-1) -- the flat padded layout makes them constant shifts;
//   * stride 2: out(ho,wo) += W[kh][kw] * X[2ho+kh-1][2wo+kw-1].  With kh-1 = 2*dh + a, kw-1 = 2*dw + b
//     (a,b in {0,1}; dh,dw in {-1,0}) this is X_ab[ho+dh][wo+dw] on the phase image X_ab[i][j] = X[2i+a][2j+b].
//     Indexed in OUTPUT geometry the phase image is again a flat matrix and the taps are constant shifts
//     dh*Wpo + dw, so one slice = one phase x 48 (or 32) channels with 1, 2, 2 or 4 taps (9 in total: no
//     zero-weight work beyond rounding each phase's K to 32).  The phase image never exists in HBM: the LDS-DMA
//     that stages the slab gathers X[2i+a][2j+b] directly (per-lane source address), pad positions of the
//     output geometry are pointed at a zero guard row.
// A block stages, per slice, the slab [p0+minoff, p0+BM+maxoff] x KS channels and the (cout tile, slice)
// weights in LDS (LDS-DMA for both), then runs the slice's K chunks on MFMA from LDS; accumulators live across
// slices; fused bias / residual / ReLU / pad-mask epilogue as in the other kernels.  LDS <= ~62 KiB -> two or
// three blocks per CU overlap each other's load and compute phases.
// KS = 48: 96-byte slab pitch, conflict-free as is.  KS = 32: 64-byte pitch, 16-byte slots XOR-swizzled by
// 2*((row>>2)&1) on the DMA source side and on the read side (conflict-free for any 16 consecutive rows).
#include "kernels.h"

namespace hrn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#define GLOBAL_AS __attribute__((address_space(1)))

__device__ __forceinline__ void tap_glds16(const GLOBAL_AS void *gsrc, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

constexpr int TAP_BM = 256;        // pixels per block (4 waves x 64)
constexpr int TAP_MAXCH = 9;       // K chunks per slice (9 taps x 32 channels)

template <int KS, int NRB>
__global__ __launch_bounds__(256, 2) void conv_tap_lds_kernel(const TapConvArgs p) {
    // LDS: [weights: max_chunks*NRB KiB][slab: slab_bytes][tap table]
    constexpr int MR = 4, ROWB = KS * 2, UPR = KS / 8, NT = 256;
    const int WMAX = p.max_chunks * NRB * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const wlds = smem;
    char *const slab = smem + WMAX;
    int *const taptab = (int *)(slab + p.slab_bytes);  // [16] LDS row offset of tap t of the current slice

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    // XCD-aware ids (block b runs on XCD b % 8): the cout tiles of one M tile are 8 ids apart
    const int nt = (blockIdx.x >> 3) % p.ntiles;
    const int mt = (blockIdx.x / (8 * p.ntiles)) * 8 + (blockIdx.x & 7);
    const int p0 = mt * TAP_BM;
    if (p0 >= p.m) return;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;

    f32x4 acc[MR][NRB];
    const int ch0 = nt * 16 * NRB + g * 4 * NRB;
#pragma unroll
    for (int j = 0; j < NRB; ++j) {
        const f32x4 b = *(const f32x4 *)(p.bias + ch0 + j * 4);
#pragma unroll
        for (int i = 0; i < MR; ++i) acc[i][j] = b;
    }

    const TapSlice *__restrict__ slices = p.slices;
    for (int s = 0; s < p.nslices; ++s) {
        const TapSlice sl = slices[s];
        if (s) __syncthreads();
        if (tid < 16) taptab[tid] = (sl.tap_off[tid < sl.ntaps ? tid : sl.ntaps - 1] - sl.minoff) * ROWB;
        // ---- weights of (slice, cout tile): linear copy of the packed image
        {
            const int wunits = sl.nchunks * NRB * 64;
            const GLOBAL_AS char *wsrc = (const GLOBAL_AS char *)sl.w + (size_t)nt * wunits * 16;
            for (int u0 = wave * 64; u0 < wunits; u0 += NT) tap_glds16(wsrc + (size_t)(u0 + lane) * 16, wlds + u0 * 16);
        }
        // ---- slab: rows [p0+minoff, p0+BM+maxoff] of the slice's view, KS channels
        {
            const int rows = TAP_BM + sl.maxoff - sl.minoff + 1;
            const int units = rows * UPR;
            const GLOBAL_AS unsigned short *src = (const GLOBAL_AS unsigned short *)sl.src;
            for (int u0 = wave * 64; u0 < units; u0 += NT) {
                int u = u0 + lane;
                if (u >= units) u = units - 1;
                const int r = u / UPR;
                int q = u - r * UPR;
                if (KS == 32) q ^= ((r >> 2) & 1) << 1;     // swizzled image: LDS slot u%4 holds source slot q
                const long i = (long)p0 + sl.minoff + r;    // row index in the (output-geometry) view
                long srow;
                if (sl.mode == 0) {
                    srow = i;                               // stride 1: the tensor itself
                } else {                                     // stride 2: gather phase (a,b)
                    // i may be negative / beyond the batch: such rows only feed masked outputs -> zero row
                    const int ii = (int)i;
                    const int n = (int)(((unsigned long long)(unsigned)(ii < 0 ? 0 : ii) * p.magic_hpwp) >> p.shift_hpwp);
                    const int rem = ii - n * p.hpwp;
                    const int ho = (int)(((unsigned long long)(unsigned)(rem < 0 ? 0 : rem) * p.magic_wp) >> p.shift_wp);
                    const int wo = rem - ho * p.wp;
                    const bool valid = ii >= 0 && ii < p.m && ho < p.h && wo < p.wd;
                    srow = valid ? (long)n * sl.src_hpwp + (long)(2 * ho + sl.a) * sl.src_wp + (2 * wo + sl.b)
                                 : -1;                      // row -1 = guard row, zeros
                }
                tap_glds16(src + srow * sl.src_c + sl.ci0 + q * 8, slab + u0 * 16);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- K chunks of this slice
        const int ktot = sl.ntaps * KS;
        for (int c = 0; c < sl.nchunks; ++c) {
            int k0 = 32 * c + 8 * g;
            if (k0 >= ktot) k0 = ktot - 8;  // zero-weight padding: any valid address
            const int t = k0 / KS, ci = k0 - t * KS;
            const int rowoff = taptab[t];
            s16x8 wf[NRB], xf[MR];
#pragma unroll
            for (int j = 0; j < NRB; ++j) wf[j] = *(const s16x8 *)(wlds + (c * NRB + j) * 1024 + lane * 16);
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                const int rb = (wave * 64 + i * 16 + li) * ROWB + rowoff;  // byte offset of the row in the slab
                int slot = ci >> 3;
                if (KS == 32) slot ^= ((rb >> 8) & 1) << 1;                // (row>>2)&1 with 64-byte rows
                xf[i] = *(const s16x8 *)(slab + rb + slot * 16);
            }
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NRB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[j]),
                                                                        __builtin_bit_cast(bf16x8, xf[i]), acc[i][j], 0,
                                                                        0, 0);
        }
    }
    (void)lds0;

    // ---- epilogue
    unsigned short *__restrict__ out = (unsigned short *)p.out;
    const unsigned short *__restrict__ res = (const unsigned short *)p.res;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int q = p0 + wave * 64 + i * 16 + li;
        if (q >= p.m) continue;
        const int n_img = (int)(((unsigned long long)(unsigned)q * p.magic_hpwp) >> p.shift_hpwp);
        const int rem = q - n_img * p.hpwp;
        const int ho = (int)(((unsigned long long)(unsigned)rem * p.magic_wp) >> p.shift_wp);
        const int wo = rem - ho * p.wp;
        const bool ok = (ho < p.h) && (wo < p.wd);
        const size_t o = (size_t)q * p.cout + ch0;
#pragma unroll
        for (int j = 0; j < NRB; ++j) {
            u32x2 r2 = {0u, 0u};
            if (res) r2 = *(const u32x2 *)(res + o + j * 4);
            float v0 = acc[i][j][0] + __uint_as_float(r2[0] << 16);
            float v1 = acc[i][j][1] + __uint_as_float(r2[0] & 0xffff0000u);
            float v2 = acc[i][j][2] + __uint_as_float(r2[1] << 16);
            float v3 = acc[i][j][3] + __uint_as_float(r2[1] & 0xffff0000u);
            if (p.relu) v0 = fmaxf(v0, 0.f), v1 = fmaxf(v1, 0.f), v2 = fmaxf(v2, 0.f), v3 = fmaxf(v3, 0.f);
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            const bf16x2 lo = {(__bf16)v0, (__bf16)v1}, hi = {(__bf16)v2, (__bf16)v3};
            u32x2 pk = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
            if (!ok) pk = u32x2{0u, 0u};
            *(u32x2 *)(out + o + j * 4) = pk;
        }
    }
}

template <int KS, int NRB>
static hipError_t launch_tap_t(const TapConvArgs &a, hipStream_t s) {
    const size_t shm = (size_t)a.max_chunks * NRB * 1024 + a.slab_bytes + 64;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)conv_tap_lds_kernel<KS, NRB>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int mtiles = (a.m + TAP_BM - 1) / TAP_BM;
    hipLaunchKernelGGL((conv_tap_lds_kernel<KS, NRB>), dim3(((mtiles + 7) / 8) * 8 * a.ntiles), dim3(256), shm, s, a);
    return hipGetLastError();
}

hipError_t launch_conv_tap_lds(const TapConvArgs &a, int ks, int nrb, hipStream_t s) {
    if (a.m <= 0) return hipSuccess;
    if (ks == 48 && nrb == 3) return launch_tap_t<48, 3>(a, s);
    if (ks == 48 && nrb == 4) return launch_tap_t<48, 4>(a, s);
    if (ks == 32 && nrb == 3) return launch_tap_t<32, 3>(a, s);
    if (ks == 32 && nrb == 4) return launch_tap_t<32, 4>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace hrn
