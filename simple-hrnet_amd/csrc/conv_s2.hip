// 3x3 / stride-2 / pad-1 convolutions with 48 (HRNet-W48, branch 0) or 32 / 64 (HRNet-W32, branches 0 / 1) input channels: the
// first convs of the fuse-down chains and the chain convs that follow them (models_/hrnet.py:36-51) as an LDS-staged implicit GEMM on bf16 MFMA, written for gfx950.
//
// Why its own kernel (DESIGN.md §5 "stride-2 slab kernel"): a stride-2 tile touches FOUR input pixels per output pixel and
// the 16 pixels of an MFMA fragment sit two input columns apart, so neither the stride-1 kernel's slab (one contiguous run
// of flat rows, 96-byte lane pitch) nor its tiling (512 pixels x 48 couts per block) carries over: the slab per output
// pixel is 4x larger and a lane pitch of 192 bytes is a 2-way bank conflict.  What pays here is the opposite split:
//   * M tile = R full output rows of one image (R*(Wo+1) flat output rows incl. the pad column).  Its input footprint --
//     virtual input rows 2*h0-1 .. 2*(h0+R-1)+1, each from column -1 to column W -- is staged ONCE per block in LDS by
//     LDS-DMA, de-interleaved by column parity on the way in: slot(vrow, plane, j) holds input column 2*j-1+plane, so the
//     16 pixels of a fragment are 16 consecutive 96-byte slots for every tap (96 = 32 mod 64 dwords*4: conflict-free, like
//     the stride-1 slab) and a tap is a constant slot shift  dh*2*Wop + (dw&1)*Wop + (dw>>1).  Round 4: an output row's two
//     virtual rows are followed by 0..7 pad slots (kernels.h: s2_pair_pad) so that a fragment that WRAPS an output row keeps
//     its 16 slot numbers consecutive mod 8 -- without them such fragments conflicted (32 % of the LDS cycles).
//   * N = ALL output channels of ALL convolutions that read this tensor at this fuse level (the 48->96 and the two 48->48
//     first convs of a stage-4 module: 192 couts), split over the waves in groups of 48: a wave keeps its group's whole
//     weight matrix (48 x 432 -> 14 K chunks x 3 fragments = 168 VGPRs) IN REGISTERS for the block's lifetime, so the K loop
//     has no weight traffic at all, no barrier, and one ds_read_b128 per three MFMAs; the slab crosses L2 -> LDS once per
//     tile for all 192 couts.
//   * two slab buffers (2 x 78 KiB): the next tile's slab lands while this one is computed; one barrier per tile.
//   * only REAL output rows are written (pad column included, as zeros); the pad row below an image's last row and the tail
//     guard rows are never touched -- they keep the zeros the buffer was allocated with (the workspace invariant stated at
//     hrn_ctx::new_tensor, ctx_plan.inc: buffers are zeroed once and reused by tensors of one geometry only).
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

constexpr int NT = 512;
#ifndef S2_DEPTH
#define S2_DEPTH 1
#endif
#ifndef S2_OPAQUE
#define S2_OPAQUE 1
#endif

__device__ __forceinline__ void glds16(const GLOBAL_AS void *gsrc, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// One configuration of the kernel: CIN input channels (a slot = CIN * 2 bytes of one input pixel, kept in LDS as CIN / 48
// sub-slots of 96 bytes in separate regions so that the lane pitch of a fragment read stays 96 bytes), NF output-channel
// fragments (16 couts each) per wave = per "part", MW pixel fragments processed together.
//   <48, 3, 2>: 14 K chunks x 3 fragments = 168 weight VGPRs, 24 accumulators            (round 3, first form)
//   <96, 2, 1>: 27 K chunks x 2 fragments = 216 weight VGPRs, 8 accumulators -- one pixel fragment at a time is what lets
//               the whole 32 x 864 weight matrix of a part stay in registers
template <int CIN, int NF, int MW>
__device__ __forceinline__ void s2_run(const GLOBAL_AS S2Problem *pp, const int ntile, const int tile0, char *smem_s2) {
    // A slot (one input pixel, CIN * 2 bytes) lives in LDS as HALVES sub-slots of ROWB bytes in separate regions, so that the
    // 16 consecutive pixels of a fragment read are ROWB bytes apart: 96 bytes (cin = 48, 96) or 32 bytes (cin = 32, 64) -- both
    // put the eight lanes of an LDS phase on disjoint banks; 64 or 128 bytes would be 2- / 4-way conflicts.
    constexpr int ROWB = s2_subslot_bytes(CIN), HALVES = CIN * 2 / ROWB, UPR = ROWB / 16;
    constexpr int NCH = (9 * CIN + 31) / 32;                         // K chunks of 32 (cin = 48: the last one half zero)
    constexpr int HALF_BYTES = s2_region_bytes(CIN);                 // LDS region of one sub-slot plane within a slab buffer
    constexpr int NSPH = (HALF_BYTES / 16 + NT - 1) / NT;            // LDS-DMA pieces per thread and region
    constexpr int NSP = NSPH * HALVES;
    constexpr int CPP = 16 * NF;                                     // couts per part
    static_assert(HALF_BYTES % 1024 == 0 && HALF_BYTES * HALVES <= kS2SlabBytes, "a region is whole LDS-DMA pieces");
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
    float *const bias_lds = (float *)(smem_s2 + 2 * kS2SlabBytes);   // [8 parts][CPP]

    // ---- LDS-DMA of a slab, one 1-KiB piece (64 lanes x 16 bytes) per call.  Piece k of a wave: half k / NSPH, units
    //      (k % NSPH) * 512 + wave * 64 ... of that half's region; unit u = sub-slot u / 6, 16-byte piece u % 6
    const int slots_per_vrow = 2 * Wop;
    const int pair_pitch = s2_pair_pitch(Wop);   // slots per output row: two virtual rows + the bank pad (kernels.h)
    struct Slab {
        const GLOBAL_AS char *src;
        char *dst;
        int units;   // per half
    };
    auto plan_slab = [&](int t, int b) {
        const int n = t / tpi, rg = t - n * tpi;
        const int h0 = rg * R;
        const int rt = Ho - h0 < R ? Ho - h0 : R;
        Slab sl;
        sl.units = (rt * pair_pitch + slots_per_vrow) * UPR;   // rt row pairs + the first virtual row of the next pair
        // first pixel of the slab: row 2*h0 - 1, column -1 of image n (guard rows / the previous image's pad row when h0 == 0)
        const long px0 = (long)n * in_hpwp + (long)(2 * h0 - 1) * in_wp - 1;
        sl.src = in + px0 * (CIN * 2);
        sl.dst = sbuf + b * kS2SlabBytes;
        return sl;
    };
    auto piece = [&](const Slab &sl, int k) {
        const int half = HALVES == 1 ? 0 : k / NSPH, kk = HALVES == 1 ? k : k - half * NSPH;
        if (kk * NT + wave * 64 < sl.units) {   // wave-uniform
            int u = kk * NT + tid;
            if (u >= sl.units) u = sl.units - 1;  // tail lanes re-read a valid unit (their LDS slots lie inside the region, unused)
            const int slot = UPR == 6 ? (int)(((unsigned)u * 43691u) >> 18) : u / UPR;  // u / 6 for u < 2^15 (UPR 2: a shift)
            const int pc = u - slot * UPR;
            // slot -> (virtual row, column parity plane, j): row pair slot / pair_pitch, then its first row, its second row or the
            // pad (those lanes fetch a valid pixel into slots nobody reads)
            const int pair = slot / pair_pitch, o = slot - pair * pair_pitch;
            const int second = o >= slots_per_vrow ? 1 : 0;
            int rem = o - second * slots_per_vrow;
            if (rem >= slots_per_vrow) rem = slots_per_vrow - 1;
            const int vrow = 2 * pair + second;
            const int plane = rem >= Wop ? 1 : 0, j = rem - plane * Wop;
            const int rel = (vrow * in_wp + 2 * j + plane) * (CIN * 2) + half * ROWB + pc * 16;
            glds16(sl.src + rel, sl.dst + half * HALF_BYTES + (kk * NT + wave * 64) * 16);
        }
    };

    {
        const Slab s0 = plan_slab(tile0, 0);
#pragma unroll
        for (int k = 0; k < NSP; ++k) piece(s0, k);
    }

    // ---- this wave's weights: NCH chunks x NF fragments, resident in registers; the biases -> LDS
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
    const int cout = pp->part[part].cout, ch0 = pp->part[part].ch0, relu = pp->part[part].relu;
    GLOBAL_AS unsigned short *const out = (GLOBAL_AS unsigned short *)pp->part[part].out;
    const float lo = relu ? 0.f : -INFINITY;   // ReLU as one v_med3 with a wave-uniform floor

    // LDS byte offset of k-group g of chunk c relative to the lane's own pixel slot (row 2*rr, plane 0, j = wo) in half 0.
    // cin = 48: chunks straddle taps (48 = 1.5 chunks), one per-lane value per chunk.  cin = 96: a tap is three whole chunks,
    // offset = tap shift (wave-uniform, computed from constants) + one of three per-lane values.
    // k-group g of chunk c covers channels ci .. ci + 7 of tap (32 c + 8 g) / CIN: sub-slot ci / (ROWB / 2), byte (ci % (ROWB / 2)) * 2.
    //   cin = 48: chunks straddle taps (48 = 1.5 chunks): one per-lane value per chunk;
    //   cin = 32 / 64 / 96: a tap is 1 / 2 / 3 whole chunks: tap shift (wave-uniform, from constants) + a per-lane value per
    //   chunk-within-tap.
    constexpr int CPT = CIN == 48 ? 1 : CIN / 32;                    // chunks per tap (cin = 48: unused)
    constexpr int NXO = CIN == 48 ? NCH : CPT;
    int xoff[NXO];
    if constexpr (CIN == 48) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            int k0 = 32 * c + 8 * g;
            if (k0 >= 9 * CIN) k0 = 0;   // zero weights: any valid slab address
            const int tap = k0 / CIN, ci = k0 - tap * CIN;
            const int dh = tap / 3, dw = tap - 3 * dh;
            xoff[c] = ((dh == 2 ? pair_pitch : dh * slots_per_vrow) + (dw & 1) * Wop + (dw >> 1)) * ROWB + ci * 2;
        }
    } else {
#pragma unroll
        for (int sub = 0; sub < CPT; ++sub) {
            const int ci = 32 * sub + 8 * g;
            xoff[sub] = (ci / (ROWB / 2)) * HALF_BYTES + (ci % (ROWB / 2)) * 2;
        }
    }
    auto chunk_off = [&](int c) -> unsigned {   // c is a compile-time constant at every call
        if constexpr (CIN == 48) {
            return (unsigned)xoff[c];
        } else {
            const int tap = c / CPT, dh = tap / 3, dw = tap - 3 * dh;
            return (unsigned)(((dh == 2 ? pair_pitch : dh * slots_per_vrow) + (dw & 1) * Wop + (dw >> 1)) * ROWB) + (unsigned)xoff[c % CPT];
        }
    };
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
        else if (nlast == 1)
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
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
            for (int f0 = wf0; f0 < mf; f0 += MW * wfs) {
                const bool last_iter = f0 + MW * wfs >= mf;     // wave-uniform
                int nfr = 0;                                     // fragments of this iteration that exist (wave-uniform)
                int tp[MW], wo[MW];
                unsigned xa[MW];
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    if (f0 + i * wfs < mf) nfr = i + 1;
                    int tt = (f0 + i * wfs) * 16 + li;
                    tp[i] = tt;
                    if (tt >= npx) tt = 0;   // dead lanes / a missing fragment: any valid pixel, never stored
                    const int rr = (int)(((unsigned long long)(unsigned)tt * magic_wop) >> shift_wop);
                    wo[i] = tt - rr * Wop;
                    xa[i] = lds0 + b * kS2SlabBytes + (rr * pair_pitch + wo[i]) * ROWB;
                }
                // cin = 32 / 64: address = (pixel slot + this lane's k-group of the chunk-within-tap) + the tap's shift.  The first sum is
                // made opaque here, or the compiler forms xoff + shift for every chunk once, outside all loops, and keeps 9 - 18
                // more address registers alive through the whole kernel (round 4: what made the fused stem kernel spill)
                unsigned xs[MW][CIN == 48 ? 1 : CPT];
                if constexpr (CIN != 48 && S2_OPAQUE) {
#pragma unroll
                    for (int i = 0; i < MW; ++i)
#pragma unroll
                        for (int sub = 0; sub < CPT; ++sub) {
                            xs[i][sub] = xa[i] + (unsigned)xoff[sub];
                            asm volatile("" : "+v"(xs[i][sub]));
                        }
                }
                auto rd_addr = [&](int i, int c) -> unsigned {
                    if constexpr (CIN == 48 || !S2_OPAQUE) {
                        return xa[i] + chunk_off(c);
                    } else {
                        const int tap = c / CPT, dh = tap / 3, dw = tap - 3 * dh;
                        return xs[i][c % CPT] + (unsigned)(((dh == 2 ? pair_pitch : dh * slots_per_vrow) + (dw & 1) * Wop + (dw >> 1)) * ROWB);
                    }
                };
                f32x4 acc[MW][NF];
#pragma unroll
                for (int i = 0; i < MW; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                // pixel fragments are requested S2_DEPTH chunks ahead (ring of S2_DEPTH + 1 register sets)
                s16x8 xf[S2_DEPTH + 1][MW];
#define S2_READ(SET, C)                                                                                         \
    {                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < MW; ++i)                                                          \
            asm volatile("ds_read_b128 %0, %1" : "=v"(xf[SET][i]) : "v"(rd_addr(i, C)));                       \
    }
#pragma unroll
                for (int c = 0; c < S2_DEPTH; ++c) S2_READ(c, c)
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const int cur = c % (S2_DEPTH + 1);
                    if (c + S2_DEPTH < NCH) {
                        S2_READ((c + S2_DEPTH) % (S2_DEPTH + 1), c + S2_DEPTH)
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(MW * S2_DEPTH) : "memory");   // chunk c landed, the next ones in flight
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(MW * (NCH - 1 - c)) : "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NF; ++j)
#pragma unroll
                        for (int i = 0; i < MW; ++i)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[c][j]),
                                                                                __builtin_bit_cast(bf16x8, xf[cur][i]), acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if ((c & 1) && pk < NSP) {   // one LDS-DMA piece every other chunk
                        piece(nx, pk);
                        ++pk;
                    }
                }
#undef S2_READ
                if (last_iter) {  // whatever is left of the next slab goes out BEFORE this wave's last stores (counted wait above)
                    for (; pk < NSP; ++pk) piece(nx, pk);
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);   // the stores below stay below: they are what the counted wait leaves in flight
                }
                // ---- epilogue: + bias, ReLU, zero at the pad column; a lane owns 4*NF contiguous channels of one pixel
                const float *bl = bias_lds + part * CPP + g * 4 * NF;
                f32x4 bs[NF];
#pragma unroll
                for (int j = 0; j < NF; ++j) bs[j] = *(const f32x4 *)(bl + 4 * j);
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    if (i >= nfr) break;
                    if (tp[i] < npx) {
                        const float hi = wo[i] < Wo ? INFINITY : 0.f;   // pad column: clamp to [0, 0]
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
                        if constexpr (NF == 3) *(GLOBAL_AS u32x2 *)(o + 8) = u32x2{pk2[4], pk2[5]};
                    }
                }
                if (last_iter) nlast = __builtin_amdgcn_readfirstlane(nfr * (NF == 3 ? 2 : 1));
            }
        }
        for (; pk < NSP; ++pk) piece(nx, pk);   // waves without fragments in this tile (and inactive ones)
    }
}

}  // namespace

__global__ __launch_bounds__(512) void conv_s2_slab_kernel(const S2Problem *__restrict__ probs, const int2 *__restrict__ map) {
    extern __shared__ __attribute__((aligned(1024))) char smem_s2[];
    const int2 e = map[blockIdx.x];
    const int prob = __builtin_amdgcn_readfirstlane(e.x & 0xff), ntile = __builtin_amdgcn_readfirstlane(e.x >> 8);
    const int tile0 = __builtin_amdgcn_readfirstlane(e.y);
    const GLOBAL_AS S2Problem *pp = (const GLOBAL_AS S2Problem *)(probs + prob);
    // <96, 2, 1> (27 chunks x 2 fragments = 216 weight VGPRs) compiles, but needs ~280 registers with everything else and
    // spills 24 of them into scratch -- whose accesses are vector-memory operations in the middle of the counted waits: the
    // 96-input-channel convolutions stay on the generic kernel (hrnet_mi355.cpp: ConvOp::s2 for cin 32 / 48 / 64)
    const int cin = pp->cin;
    if (cin == 48)
        s2_run<48, 3, 2>(pp, ntile, tile0, smem_s2);   // HRNet-W48, branch 0
    else if (cin == 32)
        s2_run<32, 2, 2>(pp, ntile, tile0, smem_s2);   // HRNet-W32, branch 0:  9 chunks x 2 fragments =  72 weight VGPRs
    else
        s2_run<64, 2, 2>(pp, ntile, tile0, smem_s2);   // HRNet-W32, branch 1: 18 chunks x 2 fragments = 144 weight VGPRs
}

hipError_t launch_conv_s2(const S2Problem *probs_dev, const void *map_dev, int nblocks, hipStream_t s) {
    if (nblocks <= 0) return hipSuccess;
    const int lds = 2 * kS2SlabBytes + 8 * 48 * 4;
    static std::atomic<unsigned long long> lds_set{0};   // per device: kernels.h set_dynamic_lds
    {
        const hipError_t e = set_dynamic_lds((const void *)conv_s2_slab_kernel, lds, lds_set);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(conv_s2_slab_kernel, dim3(nblocks), dim3(512), lds, s, probs_dev, (const int2 *)map_dev);
    return hipGetLastError();
}

}  // namespace hrn

