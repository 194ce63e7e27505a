// 3x3 / stride-1 / pad-1 convolution (the BasicBlock convs = 84 % of HRNet-W48 FLOPs) as an LDS-staged
// implicit GEMM on bf16 MFMA, written for gfx950.
//
// The flat padded NHWC layout turns the nine taps into nine constant row shifts of ONE
// activation matrix, so a block stages a single contiguous "slab" of rows
//     [p0 - (Wp+1), p0 + BM + (Wp+1))  x  KS input channels
// in LDS and serves all nine taps of that channel slice from it: the activations cross L2->LDS once per
// slice instead of nine times.  KS = 48 gives a 96-byte LDS row pitch; 96 = 32 (mod 64) makes the 16-lane
// ds_read_b128 groups (16 consecutive pixels x two 16-byte k-groups) hit all 64 banks exactly once, so the
// natural (unswizzled, unpadded) image is conflict-free and can be written by global_load_lds (LDS-DMA,
// lane-linear destination) with no staging registers.  Weights of the (cout tile, slice) arrive the same way
// from a pre-packed fragment-major image (pack_conv_lds in hrnet_mi355.cpp): a linear copy.
//
// K is flattened per slice: k = tap*KS + ci, cut into 32-wide MFMA chunks (KS=48: 432 -> 14 chunks, the last
// one half zero-padded, 3.6 % waste; the 16x16x16 MFMA that would avoid the padding issues at the same
// 16 cycles as 16x16x32 on gfx950 -- measured, tools/mfma_rate.hip).  A chunk's four 8-wide k-groups may
// belong to different taps; groups (0,1) and (2,3) never straddle a tap, which keeps the bank pattern.
//
// Tile: block = WAVES waves, wave = 16*MR pixels x 16*NRB couts (operands swapped, D = W * X^T, so a lane
// owns 4*NRB contiguous channels of one pixel -> the same wide-store epilogue as the generic kernel).
// Occupancy plan: LDS <= 80 KiB per block -> 2 blocks / CU, one loading while the other computes.
#include "kernels.h"

namespace hrn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__device__ __forceinline__ void glds16(const void *gsrc, char *lds_wave_base) {
    // 64 lanes x 16 B -> LDS [lds_wave_base + lane*16); the base must be wave-uniform
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

template <int KS, int NRB, int MR, int WAVES>
__global__ __launch_bounds__(WAVES * 64, (WAVES >= 4 ? 2 : 1)) void conv3x3_lds_kernel(const Conv3Args p) {
    constexpr int NCH = (9 * KS + 31) / 32;        // MFMA K-chunks per slice
    constexpr int ROWB = KS * 2;                   // LDS row pitch in bytes
    constexpr int UPR = KS / 8;                    // 16-byte units per slab row
    constexpr int BM = WAVES * 16 * MR;
    constexpr int WBYTES = NCH * NRB * 1024;
    constexpr int NT = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *wlds = smem;
    char *slab = smem + WBYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int nt = blockIdx.x % p.ntiles, mt = blockIdx.x / p.ntiles;
    const int p0 = mt * BM;
    const int slab_rows = BM + 2 * p.wp + 2;
    const int slab_units = slab_rows * UPR;
    const unsigned short *__restrict__ in = (const unsigned short *)p.in;

    // per-lane LDS byte offset of k-group g of chunk c, relative to the lane's own pixel row in the slab
    int xoff[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        int k0 = 32 * c + 8 * g;
        if (k0 >= 9 * KS) k0 = 0;  // zero-weight padding: any valid slab address
        const int tap = k0 / KS, ci = k0 - tap * KS;
        const int dh = tap / 3, dw = tap - 3 * dh;
        xoff[c] = (dh * p.wp + dw) * ROWB + ci * 2;
    }
    int xrow[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) xrow[i] = (wave * 16 * MR + i * 16 + li) * ROWB;

    f32x4 acc[MR][NRB];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NRB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const long slab_row0 = (long)p0 - p.wp - 1;  // first staged activation row (guard rows make it valid)
    for (int s = 0; s < p.slices; ++s) {
        if (s) __syncthreads();  // everybody done reading the previous slice
        // ---- stage weights of (cout tile nt, slice s): linear copy of the pre-packed image
        const char *wsrc = (const char *)p.w + ((size_t)nt * p.slices + s) * WBYTES;
        for (int u0 = wave * 64; u0 < WBYTES / 16; u0 += NT) glds16(wsrc + (size_t)(u0 + lane) * 16, wlds + u0 * 16);
        // ---- stage the activation slab: rows of KS channels (UPR units each) from rows of cin channels
        for (int u0 = wave * 64; u0 < slab_units; u0 += NT) {
            int u = u0 + lane;
            if (u >= slab_units) u = slab_units - 1;  // tail lanes re-read a valid unit; LDS has room for them
            const int r = u / UPR, q = u - r * UPR;
            glds16(in + (slab_row0 + r) * p.cin + s * KS + q * 8, slab + u0 * 16);
        }
        __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): LDS-DMA data of this wave has landed
        __syncthreads();
        // ---- 9 taps x KS channels as NCH chunks of K = 32
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            s16x8 wf[NRB];
#pragma unroll
            for (int j = 0; j < NRB; ++j) wf[j] = *(const s16x8 *)(wlds + (c * NRB + j) * 1024 + lane * 16);
            s16x8 xf[MR];
#pragma unroll
            for (int i = 0; i < MR; ++i) xf[i] = *(const s16x8 *)(slab + xrow[i] + xoff[c]);
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NRB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[j]),
                                                                        __builtin_bit_cast(bf16x8, xf[i]), acc[i][j], 0,
                                                                        0, 0);
        }
    }

    // ---- epilogue: + bias (+ residual) (ReLU), zero on pad pixels; lane owns 4*NRB contiguous channels
    const int ch0 = nt * 16 * NRB + g * 4 * NRB;
    float bias[4 * NRB];
#pragma unroll
    for (int c = 0; c < 4 * NRB; ++c) bias[c] = p.bias[ch0 + c];
    unsigned short *__restrict__ out = (unsigned short *)p.out;
    const unsigned short *__restrict__ res = (const unsigned short *)p.res;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
        const int q = p0 + wave * 16 * MR + i * 16 + li;
        if (q >= p.m) continue;
        const int rem = q % p.hpwp;
        const int ho = rem / p.wp, wo = rem - ho * p.wp;
        const bool ok = (ho < p.h) && (wo < p.wd);
        const size_t o = (size_t)q * p.cout + ch0;
#pragma unroll
        for (int j = 0; j < NRB; ++j) {
            s16x4 r4 = {};
            if (res) r4 = *(const s16x4 *)(res + o + j * 4);
            s16x4 o4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][j][r] + bias[j * 4 + r];
                if (res) v += bf2f((unsigned short)r4[r]);
                if (p.relu) v = fmaxf(v, 0.f);
                if (!ok) v = 0.f;
                o4[r] = (short)f2bf(v);
            }
            *(s16x4 *)(out + o + j * 4) = o4;
        }
    }
}

template <int KS, int NRB, int MR, int WAVES>
static hipError_t launch_t(const Conv3Args &a, hipStream_t s) {
    constexpr int NCH = (9 * KS + 31) / 32;
    constexpr int BM = WAVES * 16 * MR;
    const int slab_rows = BM + 2 * a.wp + 2;
    size_t shm = (size_t)NCH * NRB * 1024 + (((size_t)slab_rows * (KS / 8) + 63) / 64) * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)conv3x3_lds_kernel<KS, NRB, MR, WAVES>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int mtiles = (a.m + BM - 1) / BM;
    hipLaunchKernelGGL((conv3x3_lds_kernel<KS, NRB, MR, WAVES>), dim3(mtiles * a.ntiles), dim3(WAVES * 64), shm, s, a);
    return hipGetLastError();
}

int conv3x3_lds_block_rows(int variant) { return variant == 1 ? 256 : 256; }

hipError_t launch_conv3x3_lds(const Conv3Args &a, int ks, int nrb, int variant, hipStream_t s) {
    if (a.m <= 0) return hipSuccess;
    if (ks == 48 && nrb == 3) {
        if (variant == 1) return launch_t<48, 3, 8, 2>(a, s);   // 2 waves x 128 px
        return launch_t<48, 3, 4, 4>(a, s);                     // 4 waves x 64 px
    }
    return hipErrorInvalidValue;
}

}  // namespace hrn
