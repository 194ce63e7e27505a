// 3x3 / stride-2 / pad-1 convolutions with 48 input channels (the first convs of HRNet-W48's fuse-down chains and the
// chain convs that follow them, models_/hrnet.py:36-51) as an LDS-staged implicit GEMM on bf16 MFMA, written for gfx950.
//
// Why its own kernel (DESIGN.md §5 "stride-2 slab kernel"): a stride-2 tile touches FOUR input pixels per output pixel and
// the 16 pixels of an MFMA fragment sit two input columns apart, so neither the stride-1 kernel's slab (one contiguous run
// of flat rows, 96-byte lane pitch) nor its tiling (512 pixels x 48 couts per block) carries over: the slab per output
// pixel is 4x larger and a lane pitch of 192 bytes is a 2-way bank conflict.  What pays here is the opposite split:
//   * M tile = R full output rows of one image (R*(Wo+1) flat output rows incl. the pad column).  Its input footprint --
//     virtual input rows 2*h0-1 .. 2*(h0+R-1)+1, each from column -1 to column W -- is staged ONCE per block in LDS by
//     LDS-DMA, de-interleaved by column parity on the way in: slot(vrow, plane, j) holds input column 2*j-1+plane, so the
//     16 pixels of a fragment are 16 consecutive 96-byte slots for every tap (96 = 32 mod 64 dwords*4: conflict-free, like
//     the stride-1 slab) and a tap is a constant slot shift  dh*2*Wop + (dw&1)*Wop + (dw>>1).
//   * N = ALL output channels of ALL convolutions that read this tensor at this fuse level (the 48->96 and the two 48->48
//     first convs of a stage-4 module: 192 couts), split over the waves in groups of 48: a wave keeps its group's whole
//     weight matrix (48 x 432 -> 14 K chunks x 3 fragments = 168 VGPRs) IN REGISTERS for the block's lifetime, so the K loop
//     has no weight traffic at all, no barrier, and one ds_read_b128 per three MFMAs; the slab crosses L2 -> LDS once per
//     tile for all 192 couts.
//   * two slab buffers (2 x 78 KiB): the next tile's slab lands while this one is computed; one barrier per tile.
// K order and MFMA operand layout are those of the generic kernel (k = tap*48 + ci in 32-wide chunks, accumulators from
// zero, bias added in the epilogue), so results are BIT-IDENTICAL to conv_direct_kernel on the same convolution -- which is
// how the small-call fallback (too few tiles to fill the chip -> generic kernel) keeps a crop's result independent of the
// batch it arrives in, and how tests check this kernel element by element.
#include "kernels.h"

namespace hrn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#define GLOBAL_AS __attribute__((address_space(1)))

namespace {

constexpr int CIN = 48, ROWB = CIN * 2, UPR = CIN / 8;   // bytes and 16-byte units per slot
constexpr int NCH = 14;                                   // K = 9*48 = 432 -> 14 chunks of 32 (the last one half zero)
constexpr int NT = 512;
#ifndef S2_DEPTH
#define S2_DEPTH 1
#endif
constexpr int NSP = (kS2SlabBytes / 16 + NT - 1) / NT;    // LDS-DMA pieces per thread for a full slab

__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ unsigned pack2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }

__device__ __forceinline__ void glds16(const GLOBAL_AS void *gsrc, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

}  // namespace

__global__ __launch_bounds__(512) void conv_s2_slab_kernel(const S2Problem *__restrict__ probs, const int2 *__restrict__ map) {
    extern __shared__ __attribute__((aligned(1024))) char smem_s2[];
    const int2 e = map[blockIdx.x];
    const int prob = __builtin_amdgcn_readfirstlane(e.x & 0xff), ntile = __builtin_amdgcn_readfirstlane(e.x >> 8);
    const int tile0 = __builtin_amdgcn_readfirstlane(e.y);
    const GLOBAL_AS S2Problem *pp = (const GLOBAL_AS S2Problem *)(probs + prob);
    // the descriptor's fields as scalars, once (a field read through a pointer is re-loaded after every "memory" clobber)
    const int in_wp = pp->in_wp, in_hpwp = pp->in_hpwp, Ho = pp->ho, Wo = pp->wo, Wop = pp->wop, out_hpwp = pp->out_hpwp;
    const int R = pp->rows, tpi = pp->tiles_per_image, nparts = pp->nparts;
    const unsigned magic_wop = pp->magic_wop;
    const int shift_wop = pp->shift_wop;
    const GLOBAL_AS char *const in = (const GLOBAL_AS char *)pp->in;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    // this wave's cout group and its share of a tile's pixel fragments: fragments f0, f0 + fs, ... (host: balanced per SIMD)
    const int wpart = __builtin_amdgcn_readfirstlane((int)pp->wave_part[wave]);
    const int wf0 = __builtin_amdgcn_readfirstlane((int)pp->wave_f0[wave]), wfs = __builtin_amdgcn_readfirstlane((int)pp->wave_fs[wave]);
    const bool active = wpart < nparts;
    const int part = active ? wpart : 0;

    char *const sbuf = smem_s2;
    float *const bias_lds = (float *)(smem_s2 + 2 * kS2SlabBytes);   // [8][48]

    // ---- LDS-DMA of a slab, one 1-KiB piece (64 lanes x 16 bytes) per call; piece k of this wave = units k*512 + wave*64 ...
    const int slots_per_vrow = 2 * Wop;
    struct Slab {
        const GLOBAL_AS char *src;
        char *dst;
        int units;
    };
    auto plan_slab = [&](int t, int b) {
        const int n = t / tpi, rg = t - n * tpi;
        const int h0 = rg * R;
        const int rt = Ho - h0 < R ? Ho - h0 : R;
        Slab sl;
        sl.units = (2 * rt + 1) * slots_per_vrow * UPR;
        // first pixel of the slab: row 2*h0 - 1, column -1 of image n (guard rows / the previous image's pad row when h0 == 0)
        const long px0 = (long)n * in_hpwp + (long)(2 * h0 - 1) * in_wp - 1;
        sl.src = in + px0 * ROWB;
        sl.dst = sbuf + b * kS2SlabBytes;
        return sl;
    };
    auto piece = [&](const Slab &sl, int k) {
        if (k * NT + wave * 64 < sl.units) {   // wave-uniform
            int u = k * NT + tid;
            if (u >= sl.units) u = sl.units - 1;  // tail lanes re-read a valid unit (their LDS slots lie inside the buffer, unused)
            const int slot = (int)(((unsigned)u * 43691u) >> 18);  // u / 6 for u < 2^15
            const int pc = u - slot * UPR;
            const int vrow = slot / slots_per_vrow, rem = slot - vrow * slots_per_vrow;
            const int plane = rem >= Wop ? 1 : 0, j = rem - plane * Wop;
            const int rel = (vrow * in_wp + 2 * j + plane) * ROWB + pc * 16;
            glds16(sl.src + rel, sl.dst + (k * NT + wave * 64) * 16);
        }
    };

    {
        const Slab s0 = plan_slab(tile0, 0);
#pragma unroll
        for (int k = 0; k < NSP; ++k) piece(s0, k);
    }

    // ---- this wave's weights: 14 chunks x 3 fragments, resident in registers; its bias -> LDS
    s16x8 wf[NCH][3];
    {
        const GLOBAL_AS char *wsrc = (const GLOBAL_AS char *)pp->part[part].w + lane * 16;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int j = 0; j < 3; ++j) wf[c][j] = *(const GLOBAL_AS s16x8 *)(wsrc + (c * 3 + j) * 1024);
        if (tid < nparts * 48) {
            const int pt = tid / 48, ch = tid - pt * 48;
            bias_lds[pt * 48 + ch] = ((const GLOBAL_AS float *)pp->part[pt].bias)[pp->part[pt].ch0 + ch];
        }
    }
    const int cout = pp->part[part].cout, ch0 = pp->part[part].ch0, relu = pp->part[part].relu;
    GLOBAL_AS unsigned short *const out = (GLOBAL_AS unsigned short *)pp->part[part].out;
    const float lo = relu ? 0.f : -INFINITY;   // ReLU as one v_max with a wave-uniform floor

    // per-lane LDS byte offset of k-group g of chunk c relative to the lane's own pixel slot (row 2*rr, plane 0, j = wo)
    int xoff[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        int k0 = 32 * c + 8 * g;
        if (k0 >= 9 * CIN) k0 = 0;   // zero weights: any valid slab address
        const int tap = k0 / CIN, ci = k0 - tap * CIN;
        const int dh = tap / 3, dw = tap - 3 * dh;
        xoff[c] = (dh * slots_per_vrow + (dw & 1) * Wop + (dw >> 1)) * ROWB + ci * 2;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem_s2;

    int nlast = 0;   // stores this wave issued AFTER its last LDS-DMA piece of the previous iteration (they may stay in flight)
    for (int k = 0; k < ntile; ++k) {
        const int t = tile0 + k, b = k & 1;
        // my pieces of this tile's slab have landed.  vmcnt retires in order and counts stores: the youngest `nlast`
        // operations are the previous tile's last stores, everything older (all LDS-DMA pieces) must be complete
        if (nlast == 4)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (nlast == 2)
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // everybody's have; everybody is done reading the other buffer
        // the next tile's slab goes out piece by piece under this tile's MFMAs (an LDS-DMA instruction costs its wave
        // ~150 issue cycles; issued in one burst by all eight waves the block would compute nothing meanwhile)
        Slab nx;
        nx.units = 0, nx.src = in, nx.dst = sbuf;
        if (k + 1 < ntile) nx = plan_slab(t + 1, b ^ 1);
        int pk = 0;   // next piece to issue (wave-uniform)
        nlast = 0;

        const int n = t / tpi, rg = t - n * tpi;
        const int h0 = rg * R;
        const int rt = Ho - h0 < R ? Ho - h0 : R;
        const int npx = rt * Wop;
        const int mf = (npx + 15) >> 4;
        const long q0 = (long)n * out_hpwp + (long)h0 * Wop;   // flat output row of the tile's first pixel
        if (active) {
            for (int f0 = wf0; f0 < mf; f0 += 2 * wfs) {
                const int f1 = f0 + wfs;
                const bool two = f1 < mf;                 // wave-uniform
                const bool last_pair = f0 + 2 * wfs >= mf;
                int tp[2], wo[2];
                unsigned xa[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    int tt = (i == 0 ? f0 : f1) * 16 + li;
                    tp[i] = tt;
                    if (tt >= npx) tt = 0;   // dead lanes / the missing second fragment: any valid pixel, never stored
                    const int rr = (int)(((unsigned long long)(unsigned)tt * magic_wop) >> shift_wop);
                    wo[i] = tt - rr * Wop;
                    xa[i] = lds0 + b * kS2SlabBytes + (2 * rr * slots_per_vrow + wo[i]) * ROWB;
                }
                f32x4 acc[2][3];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                // pixel fragments are requested S2_DEPTH chunks ahead (ring of S2_DEPTH + 1 register sets)
                s16x8 xf[S2_DEPTH + 1][2];
#define S2_READ(SET, C)                                                                                   \
    {                                                                                                     \
        asm volatile("ds_read_b128 %0, %1" : "=v"(xf[SET][0]) : "v"(xa[0] + (unsigned)xoff[C]));          \
        asm volatile("ds_read_b128 %0, %1" : "=v"(xf[SET][1]) : "v"(xa[1] + (unsigned)xoff[C]));          \
    }
#pragma unroll
                for (int c = 0; c < S2_DEPTH; ++c) S2_READ(c, c)
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int cur = c % (S2_DEPTH + 1);
                    if (c + S2_DEPTH < NCH) {
                        S2_READ((c + S2_DEPTH) % (S2_DEPTH + 1), c + S2_DEPTH)
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(2 * S2_DEPTH) : "memory");   // chunk c landed, the next ones in flight
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(2 * (NCH - 1 - c)) : "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[c][j]),
                                                                            __builtin_bit_cast(bf16x8, xf[cur][0]), acc[0][j], 0, 0, 0);
                        acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[c][j]),
                                                                            __builtin_bit_cast(bf16x8, xf[cur][1]), acc[1][j], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if ((c & 1) && pk < NSP) {   // one LDS-DMA piece every other chunk
                        piece(nx, pk);
                        ++pk;
                    }
                }
#undef S2_READ
                if (last_pair) {  // whatever is left of the next slab goes out BEFORE this wave's last stores (counted wait above)
                    for (; pk < NSP; ++pk) piece(nx, pk);
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);   // the stores below stay below: they are what the counted wait leaves in flight
                }
                // ---- epilogue: + bias, ReLU, zero at the pad column; a lane owns 12 contiguous channels of one pixel
                const float *bl = bias_lds + part * 48 + g * 12;
                const f32x4 b0 = *(const f32x4 *)(bl), b1 = *(const f32x4 *)(bl + 4), b2 = *(const f32x4 *)(bl + 8);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (i == 1 && !two) break;
                    if (tp[i] < npx) {
                        const float hi = wo[i] < Wo ? INFINITY : 0.f;   // pad column: clamp to [0, 0]
                        const float lo_i = wo[i] < Wo ? lo : 0.f;
                        float v[12];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[r] = acc[i][0][r] + b0[r];
                            v[4 + r] = acc[i][1][r] + b1[r];
                            v[8 + r] = acc[i][2][r] + b2[r];
                        }
                        unsigned pk2[6];
#pragma unroll
                        for (int r = 0; r < 6; ++r) {
                            float a0, a1;
                            asm("v_med3_f32 %0, %1, %2, %3" : "=v"(a0) : "v"(v[2 * r]), "v"(lo_i), "v"(hi));
                            asm("v_med3_f32 %0, %1, %2, %3" : "=v"(a1) : "v"(v[2 * r + 1]), "v"(lo_i), "v"(hi));
                            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk2[r]) : "v"(a0), "v"(a1));
                        }
                        GLOBAL_AS unsigned short *o = out + (size_t)(q0 + tp[i]) * cout + ch0 + g * 12;
                        *(GLOBAL_AS u32x4 *)o = u32x4{pk2[0], pk2[1], pk2[2], pk2[3]};
                        *(GLOBAL_AS u32x2 *)(o + 8) = u32x2{pk2[4], pk2[5]};
                    }
                }
                if (last_pair) nlast = __builtin_amdgcn_readfirstlane(two ? 4 : 2);
            }
        }
        for (; pk < NSP; ++pk) piece(nx, pk);   // waves without fragments in this tile (and inactive ones)
    }
}

hipError_t launch_conv_s2(const S2Problem *probs_dev, const void *map_dev, int nblocks, hipStream_t s) {
    if (nblocks <= 0) return hipSuccess;
    static bool attr_set = false;
    const int lds = 2 * kS2SlabBytes + 8 * 48 * 4;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)conv_s2_slab_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(conv_s2_slab_kernel, dim3(nblocks), dim3(512), lds, s, probs_dev, (const int2 *)map_dev);
    return hipGetLastError();
}

}  // namespace hrn
