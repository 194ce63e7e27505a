// 3x3 / stride-1 / pad-1 convolution in fp32 (the parity mode; BASELINE configs[1]: HRNet-W32 256x192, batch 64, fp32) as
// an LDS-staged implicit GEMM on v_mfma_f32_16x16x4_f32 -- the fp32 sibling of conv3x3_lds.hip, sharing its problem
// descriptors, block maps and grouped launches (hrnet_mi355.cpp: Conv3Group with ks = 16).
//
// Byte for byte the geometry of the bf16 kernel's KS = 32 configuration: a slice is 16 fp32 channels = 64 bytes per slab row
// (16-byte slots XOR-swizzled by 2*((row>>2)&1), conflict-free for any 8 consecutive rows), one K chunk per tap = one
// 16-byte fragment per lane = FOUR 16x16x4 MFMAs (lane (li, g) holds k = 4g + t of its row for t = 0..3).  What differs is
// the balance: an fp32 fragment pair keeps the matrix pipe busy for 4 x 32 cycles against 16 for bf16, so LDS reads, LDS-DMA
// and the per-slice barrier are a few per cent of the MFMA time and a plain two-buffer pipeline (next slice's slab + weights
// requested one LDS-DMA piece per K chunk during a stage, vmcnt(0) + barrier at its end, fragments read one chunk ahead) is
// enough: no hand-issued reads, no counted waits.
// The generic kernel this replaces for these convolutions (conv_direct_kernel<DT_F32>) fetches every fragment from L2 and
// lives on occupancy: 0.35 of the 157 TF fp32 MFMA peak on configs[1] (BENCH_r02).
// K order: slice-major, tap, channel (the generic kernel: tap-major) -- so a convolution takes this form at EVERY batch
// size (128-pixel tiles when a launch could not fill the chip) and a crop's result does not depend on its batch.
#include "kernels.h"

namespace hrn {

typedef __attribute__((ext_vector_type(4))) float f32x4;
#define GLOBAL_AS __attribute__((address_space(1)))

namespace {

constexpr int F_ROWB = 64, F_NCH = 9, F_NT = 512;
constexpr int F_SLAB = 43008;                        // one slab buffer: 672 rows of 64 bytes
constexpr int F_MAXROWS = F_SLAB / F_ROWB;
constexpr int F_NSP = (F_SLAB / 16 + F_NT - 1) / F_NT;   // LDS-DMA pieces per thread for a slab

__device__ __forceinline__ void glds16(const GLOBAL_AS void *gsrc, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

template <int NRB, int MR>
__device__ __forceinline__ void c3f_run(const Conv3Problem &p, const int nt, const int mt0, const int tiles_this_block, const int nb,
                                        char *smem) {
    constexpr int BM = 128 * MR, WSL = F_NCH * NRB * 1024;   // weights of one (cout tile, slice)
    constexpr int NWP = (WSL / 16 + F_NT - 1) / F_NT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int m = nb * p.hpwp;
    const int mtiles = (m + BM - 1) / BM;
    int ntile = mtiles - mt0;
    if (ntile > tiles_this_block) ntile = tiles_this_block;
    if (ntile <= 0) return;
    const int S = p.slices;
    const int slab_units = (BM + 2 * p.wp + 2) * 4;
    const GLOBAL_AS float *const in = (const GLOBAL_AS float *)p.in;
    char *const wbuf = smem, *const sbuf = smem + 2 * WSL;

    // per-lane byte offset of k-group g of tap c relative to the lane's own pixel row in the slab (swizzled slot)
    int xoff[F_NCH];
#pragma unroll
    for (int c = 0; c < F_NCH; ++c) {
        const int shift = (c / 3) * p.wp + (c % 3);
        const int slot = g ^ ((((wave * 16 * MR + li + shift) >> 2) & 1) << 1);   // + 16*i rows leaves bit 2 alone
        xoff[c] = shift * F_ROWB + slot * 16;
    }
    const int xrow0 = (wave * 16 * MR + li) * F_ROWB;
    // per-lane byte offset of slab piece k's source relative to the slab's first row
    unsigned srel[F_NSP];
#pragma unroll
    for (int k = 0; k < F_NSP; ++k) {
        int u = k * F_NT + tid;
        if (u >= slab_units) u = slab_units - 1;   // tail lanes re-read a valid unit; the buffer has room for them
        const int r = u >> 2;
        const int q = (u & 3) ^ (((r >> 2) & 1) << 1);   // LDS slot u % 4 of row r holds source slot q
        srel[k] = (unsigned)(r * p.cin + q * 4) * 4u;
    }
    const int ch0 = nt * 16 * NRB + g * 4 * NRB;
    f32x4 bias[NRB];
#pragma unroll
    for (int j = 0; j < NRB; ++j) bias[j] = *(const GLOBAL_AS f32x4 *)((const GLOBAL_AS float *)p.bias + ch0 + j * 4);
    GLOBAL_AS float *const out = (GLOBAL_AS float *)p.out;
    const GLOBAL_AS float *const res = (const GLOBAL_AS float *)p.res;
    const GLOBAL_AS char *const wsrc_nt = (const GLOBAL_AS char *)p.w + (size_t)nt * S * WSL;

    // stage (tile tt, slice s) -> buffers b: the (cout tile, s) weights and the slab of channels [16 s, 16 s + 16), as
    // NWP + F_NSP one-KiB LDS-DMA pieces per wave.  A piece costs its wave ~150 issue cycles, so they go out one per K chunk
    // under the MFMAs of the stage that runs meanwhile (all at once at the stage top, both waves of every SIMD sit in DMA issue
    // together: measured 1.5 % slower)
    struct Stage {
        const GLOBAL_AS char *wsrc, *ssrc;
        char *wdst, *sdst;
        bool on;
    };
    auto plan = [&](int tt, int s, int b) {
        Stage st;
        st.on = true;
        st.wsrc = wsrc_nt + (size_t)s * WSL;
        st.wdst = wbuf + b * WSL;
        const long row0 = (long)(mt0 + tt) * BM - p.wp - 1;   // guard rows make negative / overrun rows valid
        st.ssrc = (const GLOBAL_AS char *)(in + row0 * p.cin + s * 16);
        st.sdst = sbuf + b * F_SLAB;
        return st;
    };
    auto piece = [&](const Stage &st, int k) {
        if (!st.on) return;
        if (k < NWP) {
            const int u0 = k * F_NT + wave * 64;
            if (u0 < WSL / 16) glds16(st.wsrc + (size_t)(u0 + lane) * 16, st.wdst + u0 * 16);
        } else {
            const int ks = k - NWP;
            if (ks * F_NT + wave * 64 < slab_units) glds16(st.ssrc + srel[ks], st.sdst + (ks * F_NT + wave * 64) * 16);
        }
    };
    constexpr int NPIECE = NWP + F_NSP;
    static_assert(NPIECE <= 2 * F_NCH, "at most two LDS-DMA pieces per chunk");

    {
        const Stage s0 = plan(0, 0, 0);
#pragma unroll
        for (int k = 0; k < NPIECE; ++k) piece(s0, k);
    }
    f32x4 acc[MR][NRB];
    int st = 0;
    for (int tt = 0; tt < ntile; ++tt) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
            for (int j = 0; j < NRB; ++j) acc[i][j] = bias[j];   // accumulators start at the folded-BN bias
        for (int s = 0; s < S; ++s, ++st) {
            const int b = st & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this stage's pieces (issued during the previous stage) have landed
            __syncthreads();                                    // everybody's have; everybody is done with the other buffers
            Stage nx;
            nx.on = false, nx.wsrc = wsrc_nt, nx.ssrc = (const GLOBAL_AS char *)in, nx.wdst = wbuf, nx.sdst = sbuf;
            {
                int s2 = s + 1, t2 = tt;
                if (s2 == S) s2 = 0, ++t2;
                if (t2 < ntile) nx = plan(t2, s2, b ^ 1);
            }
            const char *wl = wbuf + b * WSL + lane * 16;
            const char *xl = sbuf + b * F_SLAB + xrow0;
            // fragments one chunk ahead (two register sets)
            f32x4 wf[2][NRB], xf[2][MR];
#define C3F_READ(SET, C)                                                                              \
    {                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < NRB; ++j) wf[SET][j] = *(const f32x4 *)(wl + ((C)*NRB + j) * 1024); \
        _Pragma("unroll") for (int i = 0; i < MR; ++i) xf[SET][i] = *(const f32x4 *)(xl + xoff[C] + i * 16 * F_ROWB); \
    }
            C3F_READ(0, 0)
#pragma unroll
            for (int c = 0; c < F_NCH; ++c) {
                const int cur = c & 1;
                if (c + 1 < F_NCH) C3F_READ(cur ^ 1, c + 1)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < MR; ++i)
#pragma unroll
                        for (int j = 0; j < NRB; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[cur][j][t], xf[cur][i][t], acc[i][j], 0, 0, 0);
                piece(nx, c);
                if (c + F_NCH < NPIECE) piece(nx, c + F_NCH);
            }
#undef C3F_READ
        }
        // ---- epilogue: (+ residual) (ReLU), zero on pad pixels; a lane owns 4*NRB contiguous channels of one pixel
        const int p0 = (mt0 + tt) * BM;
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            const int q = p0 + wave * 16 * MR + i * 16 + li;   // q >= m lands in the zero tail guard: store zeros
            const int n_img = (int)(((unsigned long long)(unsigned)q * p.magic_hpwp) >> p.shift_hpwp);
            const int rem = q - n_img * p.hpwp;
            const int ho = (int)(((unsigned long long)(unsigned)rem * p.magic_wp) >> p.shift_wp);
            const int wo = rem - ho * p.wp;
            const bool ok = (q < m) && (ho < p.h) && (wo < p.wd);
            const size_t o = (size_t)q * p.cout + ch0;
#pragma unroll
            for (int j = 0; j < NRB; ++j) {
                f32x4 v = acc[i][j];
                if (res) {
                    const f32x4 r = *(const GLOBAL_AS f32x4 *)(res + (q < m ? o : (size_t)ch0) + j * 4);
                    v += r;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (p.relu) v[r] = fmaxf(v[r], 0.f);
                    if (!ok) v[r] = 0.f;
                }
                *(GLOBAL_AS f32x4 *)(out + o + j * 4) = v;
            }
        }
    }
}

template <int NRB>
__global__ __launch_bounds__(512) void conv3x3_f32_kernel(const Conv3Problem *__restrict__ probs, const int2 *__restrict__ blockmap,
                                                          const int nb) {
    extern __shared__ __attribute__((aligned(16))) char smem_f32[];
    const int2 bm = blockmap[blockIdx.x];
    // block map entry as in conv3x3_lds_kernel: x = problem | cout tile << 8 | M tiles << 16, y = first M tile | small << 30
    const Conv3Problem p = probs[bm.x & 0xff];
    const int nt = (bm.x >> 8) & 0xff, tiles = bm.x >> 16;
    const int mt0 = bm.y & 0x1fffffff;
    if (bm.y >> 30)
        c3f_run<NRB, 1>(p, nt, mt0, tiles, nb, smem_f32);
    else if (p.bm == 512)
        c3f_run<NRB, 4>(p, nt, mt0, tiles, nb, smem_f32);
    else
        c3f_run<NRB, 3>(p, nt, mt0, tiles, nb, smem_f32);
}

template <int NRB>
hipError_t launch_c3f(const Conv3Problem *probs_dev, const int2 *blockmap_dev, int nblocks, int nb, hipStream_t s) {
    constexpr int LDS = 2 * F_NCH * NRB * 1024 + 2 * F_SLAB;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static std::atomic<unsigned long long> lds_set{0};   // per device: kernels.h set_dynamic_lds
    {
        const hipError_t e = set_dynamic_lds((const void *)conv3x3_f32_kernel<NRB>, LDS, lds_set);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((conv3x3_f32_kernel<NRB>), dim3(nblocks), dim3(512), LDS, s, probs_dev, blockmap_dev, nb);
    return hipGetLastError();
}

}  // namespace

// pixels per M tile for row pitch wp: 512, or 384 when 512 rows + halo do not fit a slab buffer; 0 = unsupported
int conv3x3_f32_bm(int wp) { return 512 + 2 * wp + 2 <= F_MAXROWS ? 512 : 384 + 2 * wp + 2 <= F_MAXROWS ? 384 : 0; }

hipError_t launch_conv3x3_f32(const Conv3Problem *probs_dev, const void *blockmap_dev, int nblocks, int nb, int nrb, hipStream_t s) {
    if (nblocks <= 0) return hipSuccess;
    const int2 *bm = (const int2 *)blockmap_dev;
    if (nrb == 2) return launch_c3f<2>(probs_dev, bm, nblocks, nb, s);
    if (nrb == 3) return launch_c3f<3>(probs_dev, bm, nblocks, nb, s);
    return hipErrorInvalidValue;
}

}  // namespace hrn
