// 3x3 / stride-1 / pad-1 convolution (the BasicBlock convs = 84 % of HRNet-W48 FLOPs) as an LDS-staged,
// software-pipelined implicit GEMM on bf16 MFMA, written for gfx950.
//
// Data movement
//   * The flat padded NHWC layout turns the nine taps into nine constant row shifts of ONE
//     activation matrix, so a block stages a single contiguous "slab" of rows
//         [p0 - (Wp+1), p0 + BM + (Wp+1))  x  KS input channels
//     in LDS and serves all nine taps of that channel slice from it (activations cross L2->LDS once per
//     slice, not nine times).  KS = 48 -> 96-byte LDS row pitch; 96 = 32 (mod 64) makes each 16-lane
//     ds_read_b128 group (16 consecutive pixels x two 16-byte k-groups) cover all 64 banks exactly once, so
//     the natural, unswizzled image is conflict-free and is written by global_load_lds (LDS-DMA) directly.
//   * Weights of (cout tile, slice) come from a pre-packed fragment-major image (pack_conv_lds in
//     hrnet_mi355.cpp): a linear LDS-DMA copy, read back with lane*16 addressing.
//   * K is flattened per slice, k = tap*KS + ci, in 32-wide MFMA chunks: 432 -> 14 chunks (last one half
//     zero; the 16x16x16 MFMA that would avoid the padding costs the same 16 cycles on gfx950 -- measured).
// Pipeline (one block per CU, 8 waves = 2 per SIMD, persistent over `tiles_per_block` M tiles)
//   * unit of work = half a slice (7 chunks).  LDS holds two weight half-buffers (2 x 21 KiB) and two slab
//     buffers (2 x 56 KiB).  At the top of half-stage h every wave waits for its own LDS-DMA (vmcnt 0), the
//     block meets at ONE barrier, the loads of half-stage h+1 (next weight half; next slab when a new slice
//     or tile starts) are issued, then half-stage h is computed -- so every load has a full compute phase
//     (>= 1344 MFMA cycles) to land, also across tile boundaries and under the epilogue.
//   * single-slice problems (cin == 48) keep both weight halves resident across tiles.
//   * an LDS-DMA instruction costs its wave ~100+ issue cycles and the epilogue is store-latency bound, so
//     two waves share each SIMD: while one issues loads / stores, the other keeps the MFMA pipe busy.
//     The residual tile is requested before the last half-stage's MFMAs and is in registers by the epilogue.
// Tile: wave = 16*MR pixels x 48 couts (MR = 4: BM = 512, MR = 3: BM = 384 for the 96x72 branch whose halo
// would not fit twice); operands swapped (D = W * X^T) so a lane owns 12 contiguous channels of one pixel.
// One launch covers a GROUP of independent convolutions (the k-th conv of every branch of a stage module).
#include "kernels.h"

namespace hrn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// The problem descriptors are read from memory, so their pointers are generic; tell the compiler they are
// global, otherwise every access becomes a FLAT op, which counts on lgkmcnt as well and forces lgkmcnt(0)
// drains in front of the MFMAs (measured: every wait in the chunk loop was a full drain).
#define GLOBAL_AS __attribute__((address_space(1)))
typedef const GLOBAL_AS unsigned short *gcu16;
typedef GLOBAL_AS unsigned short *gu16;

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// ReLU as ONE v_max_f32: fmaxf() makes hipcc emit a canonicalising v_max in front of the real one
__device__ __forceinline__ float relu1(float x) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

__device__ __forceinline__ void glds16(const GLOBAL_AS void *gsrc, char *lds_wave_base) {
    // 64 lanes x 16 B -> LDS [lds_wave_base + lane*16); the base must be wave-uniform
    __builtin_amdgcn_global_load_lds(gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// Kernel configurations.  KS = 48 (HRNet-W48 branches): 96-byte slab pitch, a slice = 14 chunks in two parts of 7.
// KS = 32 (HRNet-W32 branches, layer1's 64->64, transition1's 256->48): 64-byte pitch with the 16-byte slots
// XOR-swizzled by 2*((row>>2)&1) (conflict-free for any 16 consecutive rows), a slice = 9 chunks (one per tap, no
// K padding) in a single part.  LDS = 2 weight part buffers + 2 slab buffers <= 160 KiB.
template <int KS_, int NRB_>
struct C3Cfg {
    static constexpr int KS = KS_, NRB = NRB_;
    static constexpr int PARTS = KS == 48 ? 2 : 1;            // parts per slice
    static constexpr int CPP = KS == 48 ? 7 : 9;              // chunks per part
    static constexpr int NCH = PARTS * CPP;                   // chunks per slice
    static constexpr int WPART = CPP * NRB * 1024;            // bytes of one weight part
    static constexpr int SLAB = KS == 48 ? 57344 : 43008;     // one slab buffer
    static constexpr int LDS = 2 * WPART + 2 * SLAB;
    static constexpr int ROWB = KS * 2;
    static constexpr int MAXROWS = SLAB / ROWB;               // slab rows that fit
    static constexpr int NWP = (WPART / 16 + 511) / 512;      // LDS-DMA pieces per wave for one weight part
    static constexpr int NSP = (SLAB / 16 + 511) / 512;       // ... for one slab
    static_assert(LDS <= 160 * 1024 - 256, "LDS budget");
};

#ifdef HRN_Q_TIMING   // debug (tools/cu_timeline.py): per block -- which CU ran it, from when to when (s_memrealtime, 100 MHz), of the last four launches
__device__ long long *g_q_timing = nullptr;
__device__ int g_q_seq = 0;
__global__ void q_seq_bump() { ++g_q_seq; }
constexpr int kQSlots = 4, kQBlocks = 8192;
#endif
#ifdef HRN_C3_TIMING
#define C3_T(x) const long long x = __builtin_amdgcn_s_memtime()
__device__ long long *g_c3_timing = nullptr;
#else
#define C3_T(x)
#endif

template <class CFG, int MR>
__device__ __forceinline__ void conv3_run(const Conv3Problem &p, const int nt, const int mt0, const int tiles_this_block,
                                          const int nb, char *smem) {
    constexpr int KS = CFG::KS, NRB = CFG::NRB, ROWB = CFG::ROWB, UPR = KS / 8, NT = 512;
    constexpr int BM = 128 * MR;
    constexpr int PARTS = CFG::PARTS, CPP = CFG::CPP, NCH = CFG::NCH, WPART = CFG::WPART, SLABB = CFG::SLAB;
    constexpr int NWP = CFG::NWP, SLAB_ITERS = CFG::NSP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar branches, SALU M0
    const int li = lane & 15, g = lane >> 4;
    // waves 4-7 are the younger wave of each SIMD and lose every issue arbitration to their partner (priority,
    // then age): static priority for them evens the two out (cdna_hip_programming.md T5, static form)
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    const int m = nb * p.hpwp;
    const int mtiles = (m + BM - 1) / BM;
    int ntile = mtiles - mt0;
    if (ntile > tiles_this_block) ntile = tiles_this_block;
    if (ntile <= 0) return;
    const int S = p.slices;
    const int slab_units = (BM + 2 * p.wp + 2) * UPR;
    const gcu16 in = (gcu16)p.in;
    char *const wbuf = smem;
    char *const sbuf = smem + 2 * WPART;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;  // LDS byte address

    // per-lane LDS byte offset of k-group g of chunk c, relative to the lane's own pixel row in the slab
    int xoff[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        int k0 = 32 * c + 8 * g;
        if (k0 >= 9 * KS) k0 = 0;  // zero-weight padding: any valid slab address
        const int tap = k0 / KS, ci = k0 - tap * KS;
        const int dh = tap / 3, dw = tap - 3 * dh;
        const int shift = dh * p.wp + dw;
        int slot = ci >> 3;
        if (KS == 32) slot ^= (((wave * 16 * MR + li + shift) >> 2) & 1) << 1;  // + 16*i rows leaves bit 2 alone
        xoff[c] = shift * ROWB + slot * 16;
    }
    const int xrow0 = (wave * 16 * MR + li) * ROWB;
    // per-lane element offset of the k-th slab LDS-DMA piece relative to the slab's first row
    unsigned srel[SLAB_ITERS];
#pragma unroll
    for (int k = 0; k < SLAB_ITERS; ++k) {
        int u = k * NT + tid;
        if (u >= slab_units) u = slab_units - 1;  // tail lanes re-read a valid unit; LDS has room for them
        const int r = u / UPR;
        int q = u - r * UPR;
        if (KS == 32) q ^= ((r >> 2) & 1) << 1;  // swizzled image: LDS slot u % 4 of row r holds source slot q
        srel[k] = (unsigned)(r * p.cin + q * 8) * 2u;  // bytes
    }

    const int ch0 = nt * 16 * NRB + g * 4 * NRB;
    float bias[4 * NRB];
#pragma unroll
    for (int c = 0; c < 4 * NRB; ++c) bias[c] = ((const GLOBAL_AS float *)p.bias)[ch0 + c];
    const gu16 out = (gu16)p.out;
    const gcu16 res = (gcu16)p.res;
    const bool has_res = p.res != nullptr;
    const GLOBAL_AS char *const wsrc_nt = (const GLOBAL_AS char *)p.w + (size_t)nt * S * (PARTS * WPART);

    // The LDS-DMA of half-stage (tt, s, hf) is cut into per-wave "pieces" (one 1 KiB instruction each):
    // pieces 0..2 = this wave's share of the weight half -> wbuf[hf]; pieces 3..9 = its share of the slab of
    // (tile tt, slice s) -> sbuf[par] (only when hf == 0).  A piece costs its wave ~150 issue cycles, so they
    // are spread over the chunk loop of the half-stage that runs meanwhile (the SIMD partner's MFMAs cover it).
#ifdef HRN_C3_NODMA
    const bool tt_guard = nb > 0;
#endif
    // piece counts of THIS wave (wave-uniform; plain scalars -- a counter bumped inside the lambdas ends up in
    // scratch memory, and every scratch access is a VMEM op that drains the LDS-DMA queue with vmcnt(0))
    int nw_wave = (WPART / 16 - wave * 64 + NT - 1) / NT;         // weight pieces: k*512 + wave*64 < WPART/16
    nw_wave = nw_wave < 0 ? 0 : (nw_wave > NWP ? NWP : nw_wave);
    int ns_wave = (slab_units - wave * 64 + NT - 1) / NT;       // slab pieces: k*512 + wave*64 < slab_units
    ns_wave = ns_wave < 0 ? 0 : (ns_wave > SLAB_ITERS ? SLAB_ITERS : ns_wave);
    int npost = 0;  // LDS-DMA instructions issued after the residual request (last half-stage of a tile)
    int nslab = 0;  // slab pieces issued during the hf == 0 half-stage (they may stay in flight one more)
    struct Next {
        const GLOBAL_AS char *wsrc;   // nullptr: weights stay resident
        char *wdst;
        const GLOBAL_AS char *ssrc;   // nullptr: no slab in this half-stage
        char *sdst;
    };
    // weights of part (tt, s, part) -> wbuf[buf]   (single-slice problems keep all their parts resident)
    auto plan_w = [&](Next &n, int tt, int s, int part, int buf) {
        n.wsrc = (S > 1 || tt == 0) ? wsrc_nt + (size_t)(PARTS * s + part) * WPART : nullptr;
        n.wdst = wbuf + buf * WPART;
    };
    // slab of (tile tt, slice s) -> sbuf[par]
    auto plan_s = [&](Next &n, int tt, int s, int par) {
        const long row0 = (long)(mt0 + tt) * BM - p.wp - 1;  // guard rows make negative / overrun rows valid
        n.ssrc = (const GLOBAL_AS char *)(in + row0 * p.cin + s * KS);
        n.sdst = sbuf + par * SLABB + wave * 1024;
    };
    auto piece = [&](const Next &n, int idx) {
#ifdef HRN_C3_NODMA  // ablation build (tools/c3_timing.py): results are garbage, only the timing is of interest
        if (tt_guard) return;
#endif
        if (idx < NWP) {
            const int u0 = idx * NT + wave * 64;
            if (n.wsrc && u0 < WPART / 16) {
                glds16(n.wsrc + (size_t)(u0 + lane) * 16, n.wdst + u0 * 16);
            }
        } else {
            const int k = idx - NWP;
            if (n.ssrc && k * NT + wave * 64 < slab_units) {
                glds16(n.ssrc + srel[k], n.sdst + k * NT * 16);
            }
        }
    };
    constexpr int NPIECE = NWP + SLAB_ITERS;
    static_assert(NPIECE <= 2 * CPP, "at most two LDS-DMA pieces per chunk");
    // LDS-DMA schedule.  A piece costs its wave ~150 issue cycles during which it issues no MFMA, so the pieces are
    // spread thin: at most two per chunk, over every chunk of every half-stage.  (Bunching them -- three per chunk
    // in four chunks, the two waves of a SIMD in disjoint chunk ranges -- was 2 % slower: a wave that sits in ~450
    // cycles of DMA issue lets its partner run a whole chunk ahead and the pair drifts apart until the barrier.)
    // Two-part slices: the weights of part 1 must go out during part 0 and those of the next part 0 during part 1
    // (two weight buffers).  The next slab could go out any time during the stage; K0 of its pieces go behind the
    // weights in part 0, the rest ahead of the weights in part 1.  Measured: K0 = all (12 pieces in part 0, 3 in
    // part 1) beats the balanced 8 / 7 split by 1 % -- slab pieces issued in part 1 are still in flight at the next
    // stage's barrier.
#ifndef HRN_C3_K0
#define HRN_C3_K0 SLAB_ITERS
#endif
    constexpr int K0 = PARTS == 2 ? (HRN_C3_K0) : SLAB_ITERS;

    f32x4 acc[MR][NRB];
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    constexpr int N16 = NRB / 2;       // 16-byte pieces of the lane's 4*NRB contiguous channels, plus 8 bytes if NRB is odd
    u32x4 rpre4[MR][N16 ? N16 : 1];
    u32x2 rpre2[MR];
    int wcount = 0;  // parts executed so far (selects the weight buffer when a slice is a single part)
    bool after_epilogue = false;
    int slab_par = 0;
#ifdef HRN_C3_TIMING
    long long t_wait = 0, t_issue = 0, t_comp = 0, t_epi = 0;
    int n_half = 0;
    C3_T(t_begin);
#endif
    {
        Next n0;
        plan_w(n0, 0, 0, 0, 0);
        plan_s(n0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NPIECE; ++k) piece(n0, k);
    }
    for (int tt = 0; tt < ntile; ++tt) {
#pragma unroll
        for (int i = 0; i < MR; ++i)  // accumulators start at the folded-BN bias
#pragma unroll
            for (int j = 0; j < NRB; ++j) acc[i][j] = f32x4{bias[j * 4], bias[j * 4 + 1], bias[j * 4 + 2], bias[j * 4 + 3]};
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int hf = 0; hf < PARTS; ++hf) {  // hf = part of the slice
                C3_T(tA);
                // this wave's LDS-DMA for this half-stage has landed.  vmcnt retires in order and counts stores:
                // right after an epilogue the youngest 2*MR operations are its stores, which may stay in flight
                // (a single-slice problem keeps both weight halves resident after its first tile: its second half-stage
                // waits for nothing and reads no buffer that is being refilled -- no wait, no barrier, no pipeline refill
                // in lock-step)
                const bool resident = PARTS == 2 && hf == 1 && S == 1 && tt > 0;
                if (resident) {
                } else if (PARTS == 2 && hf == 1) {
                    // the youngest `nslab` operations are the NEXT slice's slab pieces (issued during hf == 0,
                    // after this half-stage's weights): they get a second half-stage to land
                    switch (nslab) {
                        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                        default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                    }
                } else if (after_epilogue) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(MR * (N16 + (NRB & 1))) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                after_epilogue = false;
                if (!resident) __builtin_amdgcn_s_barrier();  // everyone's has; everyone is done reading the buffers refilled below
                C3_T(tB);
                // ---- what to prefetch while this half-stage computes (issued piecewise inside the chunk loop):
                //      hf == 0: the weights of (s, hf 1) and already the slab of the NEXT slice / tile;
                //      hf == 1: the weights of the next slice's first half.
                Next nx;
                nx.wsrc = nullptr, nx.ssrc = nullptr, nx.wdst = wbuf, nx.sdst = sbuf;
                // weight buffer of the current part: two-part slices alternate by part; one-part slices alternate by
                // a running count (a single-slice, single-part problem keeps its weights in buffer 0)
                const int wcur = PARTS == 2 ? hf : (S == 1 ? 0 : (wcount & 1));
                {
                    int s2 = s + 1, t2 = tt;
                    if (s2 == S) s2 = 0, ++t2;
                    if (PARTS == 2) {
                        if (hf == 0) {
                            plan_w(nx, tt, s, 1, 1);
                            if (t2 < ntile) plan_s(nx, t2, s2, slab_par ^ 1);
                            nslab = nx.ssrc ? (ns_wave < K0 ? ns_wave : K0) : 0;
                        } else {
                            if (t2 < ntile) plan_w(nx, t2, s2, 0, 0);
                            if (t2 < ntile) plan_s(nx, t2, s2, slab_par ^ 1);  // the pieces part 0 left over
                            npost = (nx.wsrc ? nw_wave : 0) + (nx.ssrc && ns_wave > K0 ? ns_wave - K0 : 0);
                        }
                    } else if (t2 < ntile) {  // one part per slice: next slice's weights and slab together
                        plan_w(nx, t2, s2, 0, S == 1 ? 0 : (wcur ^ 1));
                        plan_s(nx, t2, s2, slab_par ^ 1);
                        npost = (nx.wsrc ? nw_wave : 0) + ns_wave;
                    } else {
                        npost = 0;
                    }
                }
                // ---- last half-stage of the tile: request the residual tile now, it lands under the MFMAs
                if (hf == PARTS - 1 && s == S - 1) {
                    const int p0r = (mt0 + tt) * BM + wave * 16 * MR + li;
                    if (has_res) {
#pragma unroll
                        for (int i = 0; i < MR; ++i) {
                            int q = p0r + i * 16;
                            if (q >= m) q = 0;
                            const gcu16 rp = res + (size_t)q * p.cout + ch0;
                            // hand-issued loads: waited for with a COUNTED vmcnt in the epilogue
#pragma unroll
                            for (int v = 0; v < N16; ++v)
                                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(rpre4[i][v]) : "v"(rp), "i"(v * 16));
                            if (NRB & 1)
                                asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(rpre2[i]) : "v"(rp), "i"(N16 * 16));
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < MR; ++i) {
#pragma unroll
                            for (int v = 0; v < (N16 ? N16 : 1); ++v) rpre4[i][v] = u32x4{0u, 0u, 0u, 0u};
                            rpre2[i] = u32x2{0u, 0u};
                        }
                    }
                }
                C3_T(tC);
                // ---- compute the CPP chunks of K = 32 of this part from wbuf[wcur] and the current slab
                // Fragment reads are issued by hand (inline asm) one chunk ahead, with COUNTED waits: hipcc would
                // drain lgkmcnt(0) in front of every other MFMA block here, stalling on reads it has just issued.
                // Order is pinned with sched_barrier(0) (an MFMA must not be hoisted above the wait that covers
                // its operands; cdna_hip_programming.md rule 18).
                s16x8 wf[2][NRB], xf[2][MR];
                const unsigned wl_a = lds0 + wcur * WPART + lane * 16;
                const unsigned sl_a = lds0 + 2 * WPART + slab_par * SLABB + xrow0;
#define C3_READ_CHUNK(SET, C)                                                                                  \
    {                                                                                                          \
        _Pragma("unroll") for (int j = 0; j < NRB; ++j)                                                        \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[SET][j]) : "v"(wl_a), "i"(((C)*NRB + j) * 1024)); \
        const unsigned xa = sl_a + xoff[hf * CPP + (C)];                                                    \
        _Pragma("unroll") for (int i = 0; i < MR; ++i)                                                         \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[SET][i]) : "v"(xa), "i"(i * 16 * ROWB));    \
    }
                C3_READ_CHUNK(0, 0)
#pragma unroll
                for (int c = 0; c < CPP; ++c) {
                    const int cur = c & 1, nxt = cur ^ 1;
                    if (c + 1 < CPP) {
                        C3_READ_CHUNK(nxt, c + 1)
                    }
                    {
                        // (calls written out: as a loop hipcc spills 18 VGPRs to scratch here)
#define C3_ITEM(T)                                                                                     \
    {                                                                                                  \
        const int t_ = (T);                                                                            \
        if (PARTS == 2 && hf == 1) {                                                                   \
            if (t_ < SLAB_ITERS - K0)                                                                  \
                piece(nx, NWP + K0 + t_);                                                              \
            else if (t_ < SLAB_ITERS - K0 + NWP)                                                       \
                piece(nx, t_ - (SLAB_ITERS - K0));                                                     \
        } else if (t_ < (PARTS == 2 ? NWP + K0 : NPIECE)) {                                            \
            piece(nx, t_);                                                                             \
        }                                                                                              \
    }
                        const int n_items = PARTS == 2 ? (hf == 0 ? NWP + K0 : SLAB_ITERS - K0 + NWP) : NPIECE;
                        const int extra = n_items > CPP ? n_items - CPP : 0;  // that many chunks carry two pieces
                        if (c < extra) {
                            C3_ITEM(2 * c)
                            C3_ITEM(2 * c + 1)
                        } else {
                            C3_ITEM(c + extra)
                        }
#undef C3_ITEM
                    }
                    if (c + 1 < CPP)
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(NRB + MR) : "memory");  // chunk c landed, c+1 in flight
                    else
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < MR; ++i)
#pragma unroll
                        for (int j = 0; j < NRB; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, wf[cur][j]), __builtin_bit_cast(bf16x8, xf[cur][i]),
                                acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef C3_READ_CHUNK
#ifdef HRN_C3_TIMING
                C3_T(tD);
                t_wait += tB - tA, t_issue += tC - tB, t_comp += tD - tC, ++n_half;
#endif
                ++wcount;
                if (hf == PARTS - 1) slab_par ^= 1;
            }
        }
        // ---- epilogue: + bias (+ residual) (ReLU), zero on pad pixels; lane owns 12 contiguous channels
        C3_T(tE);
        // the residual loads are older than the `npost` LDS-DMA instructions issued after them: wait for exactly those
        if (has_res) {
            switch (npost) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;  // stricter than needed: safe
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        const int p0 = (mt0 + tt) * BM;
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            const int q = p0 + wave * 16 * MR + i * 16 + li;  // q >= m lands in the zero tail guard: store zeros
            const int n_img = (int)(((unsigned long long)(unsigned)q * p.magic_hpwp) >> p.shift_hpwp);
            const int rem = q - n_img * p.hpwp;
            const int ho = (int)(((unsigned long long)(unsigned)rem * p.magic_wp) >> p.shift_wp);
            const int wo = rem - ho * p.wp;
            const bool ok = (q < m) && (ho < p.h) && (wo < p.wd);
            const size_t o = (size_t)q * p.cout + ch0;
            unsigned pk[2 * NRB];
#pragma unroll
            for (int j = 0; j < NRB; ++j) {
                // residual: two bf16 per dword -> fp32 with one shift / one mask each
                const unsigned r01 = (j >> 1) < N16 ? rpre4[i][(j >> 1) < N16 ? (j >> 1) : 0][2 * (j & 1)] : rpre2[i][0];
                const unsigned r23 = (j >> 1) < N16 ? rpre4[i][(j >> 1) < N16 ? (j >> 1) : 0][2 * (j & 1) + 1] : rpre2[i][1];
                float v0 = acc[i][j][0] + __uint_as_float(r01 << 16);
                float v1 = acc[i][j][1] + __uint_as_float(r01 & 0xffff0000u);
                float v2 = acc[i][j][2] + __uint_as_float(r23 << 16);
                float v3 = acc[i][j][3] + __uint_as_float(r23 & 0xffff0000u);
                if (p.relu) v0 = relu1(v0), v1 = relu1(v1), v2 = relu1(v2), v3 = relu1(v3);
                typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                const bf16x2 lo = {(__bf16)v0, (__bf16)v1}, hi = {(__bf16)v2, (__bf16)v3};  // RNE, v_cvt_pk_bf16_f32
                pk[2 * j] = ok ? __builtin_bit_cast(unsigned, lo) : 0u;
                pk[2 * j + 1] = ok ? __builtin_bit_cast(unsigned, hi) : 0u;
            }
            // 8*NRB contiguous bytes per lane in 16-byte stores (+ one 8-byte store when NRB is odd): the store
            // issue count is what the tail costs
#pragma unroll
            for (int v = 0; v < N16; ++v)
                *(GLOBAL_AS u32x4 *)(out + o + v * 8) = u32x4{pk[4 * v], pk[4 * v + 1], pk[4 * v + 2], pk[4 * v + 3]};
            if (NRB & 1) *(GLOBAL_AS u32x2 *)(out + o + N16 * 8) = u32x2{pk[4 * N16], pk[4 * N16 + 1]};
        }
        after_epilogue = true;
#ifdef HRN_C3_TIMING
        C3_T(tF);
        t_epi += tF - tE;
#endif
    }
#if defined(HRN_C3_TIMING) && !defined(HRN_C3_TIMING_FUSED_ONLY)
    if (lane == 0 && g_c3_timing) {
        C3_T(t_end);
        long long *o = g_c3_timing + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[0] = t_wait, o[1] = t_issue, o[2] = t_comp, o[3] = t_epi, o[4] = t_end - t_begin, o[5] = n_half, o[6] = MR,
        o[7] = S;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// A whole BasicBlock of the 48-channel branch in one pass (modules.py:56-72: conv1+BN+ReLU, conv2+BN, + x, ReLU).
// At 144-216 FLOP per HBM byte the two convolutions are bandwidth- and vector-memory-issue-bound when run one after
// the other; fused, the intermediate never leaves the CU and the residual is already there:
//   LDS = W1 | W2 (2 x 42 KiB, resident for the block's whole life) | XY (76 KiB)
//   per tile of BM = 512 output pixels (flat rows [p0, p0 + 512)), halo = wp + 1:
//     X  = input rows [p0 - 2 halo, p0 + 512 + 2 halo)                    -> XY            (LDS-DMA, 10 pieces per wave)
//     C1 : Y = relu(W1 * X + b1), zero on pad pixels, rows [p0 - halo, p0 + 512 + halo), as bf16
//          (its 41-42 pixel fragments are dealt 5-6 per wave; the last one is pulled back to end on the last row);
//          the lane's residual values (X centre rows) are read into registers, then Y overwrites X in place
//     C2 : Z = relu(W2 * Y + b2 + X) for rows [p0, p0 + 512)               -> global
//   and the next tile's X is requested before the epilogue's stores, which it lands under.
// Bit-identical to the two separate launches (same K order, same bf16 rounding of Y, same epilogue arithmetic).
// The vector-memory instructions per wave and 2 x 512 convolved pixels drop from ~43 to 18, HBM traffic from five
// tensor passes to two; the price is 1 + 2 halo / 512 = 1.29 x the MFMAs in C1 (1.145 x overall at wp = 73).
constexpr int BBF_W = 14 * 3 * 1024;            // one packed weight image (cout tile 0, slice 0, both parts)
constexpr int BBF_XY = 4864 * 16;               // 810 rows of 96 B, rounded up to whole 64-lane pieces
constexpr int BBF_LDS = 2 * BBF_W + BBF_XY;     // = 160 KiB
static_assert(BBF_LDS <= 160 * 1024, "LDS budget");

template <int NF>
__device__ __forceinline__ void bbf_conv1(const Conv3Problem &p, const int (&xoff)[14], const int bvec, const unsigned lds0,
                                          const int row_first, const int row_last, const long q_first, const int m,
                                          const int lane, const unsigned res_a,
                                          __attribute__((ext_vector_type(2))) unsigned (&rpre)[4][3], long long &t_loop) {
    constexpr int NRB = 3, NCH = 14, ROWB = 96;
    const int li = lane & 15, g = lane >> 4;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    f32x4 acc[NF][NRB];
    {
        float bias[4 * NRB];
#pragma unroll
        for (int c = 0; c < 4 * NRB; ++c) bias[c] = __int_as_float(__builtin_amdgcn_ds_bpermute((g * 4 * NRB + c) * 4, bvec));
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < NRB; ++j) acc[i][j] = f32x4{bias[j * 4], bias[j * 4 + 1], bias[j * 4 + 2], bias[j * 4 + 3]};
    }
    // which of this lane's NF pixels are real (not pad, inside [0, m)): worked out ahead of the loop, where the VALU
    // work hides under the SIMD partner's MFMAs, instead of in the tail everybody waits for
    unsigned okbits = 0;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int yr = (i == NF - 1 ? row_last : row_first + i * 16) + li;
        const long q = q_first + yr;
        const unsigned uq = (unsigned)q;
        const int n_img = (int)(((unsigned long long)uq * p.magic_hpwp) >> p.shift_hpwp);
        const int rem = (int)uq - n_img * p.hpwp;
        const int ho = (int)(((unsigned long long)(unsigned)rem * p.magic_wp) >> p.shift_wp);
        const int wo = rem - ho * p.wp;
        okbits |= (q >= 0 && q < m && ho < p.h && wo < p.wd) ? 1u << i : 0u;
    }
    s16x8 wf[2][NRB], xf[2][NF];
    const unsigned wl_a = lds0 + lane * 16;
    unsigned sl_a = lds0 + 2 * BBF_W + (row_first + li) * ROWB;
    unsigned sl_z = lds0 + 2 * BBF_W + (row_last + li) * ROWB;   // the wave's last fragment (may be pulled back)
    // (opaque per call: otherwise the 28 per-chunk addresses sl + xoff[c] are hoisted out of the tile loop and held in
    // registers across it -- the registers the next tile's X needs during conv2)
    asm volatile("" : "+v"(sl_a), "+v"(sl_z));
#define BBF_READ1(SET, C)                                                                                         \
    {                                                                                                             \
        _Pragma("unroll") for (int j = 0; j < NRB; ++j)                                                           \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[SET][j]) : "v"(wl_a), "i"(((C)*NRB + j) * 1024)); \
        const unsigned xa = sl_a + xoff[C], xz = sl_z + xoff[C];                                                   \
        _Pragma("unroll") for (int i = 0; i < NF - 1; ++i)                                                        \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[SET][i]) : "v"(xa), "i"(i * 16 * ROWB));       \
        asm volatile("ds_read_b128 %0, %1" : "=v"(xf[SET][NF - 1]) : "v"(xz));                                     \
    }
    BBF_READ1(0, 0)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int cur = c & 1, nxt = cur ^ 1;
        if (c + 1 < NCH) {
            BBF_READ1(nxt, c + 1)
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(NRB + NF) : "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < NRB; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[cur][j]),
                                                                    __builtin_bit_cast(bf16x8, xf[cur][i]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef BBF_READ1
#ifdef HRN_C3_TIMING
    t_loop = __builtin_amdgcn_s_memtime();
#endif
    // the residual of this lane's conv2 pixels = X centre rows, fetched before Y overwrites them
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int v = 0; v < 3; ++v)
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(rpre[i][v]) : "v"(res_a), "i"(i * 16 * ROWB + v * 8));
    // ReLU, zero on pad pixels and outside [0, m), bf16: the values the separate conv1 launch would have stored
    unsigned pk[NF][2 * NRB];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const bool ok = (okbits >> i) & 1u;
#pragma unroll
        for (int j = 0; j < NRB; ++j) {
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            const bf16x2 lo = {(__bf16)relu1(acc[i][j][0]), (__bf16)relu1(acc[i][j][1])};
            const bf16x2 hi = {(__bf16)relu1(acc[i][j][2]), (__bf16)relu1(acc[i][j][3])};
            pk[i][2 * j] = ok ? __builtin_bit_cast(unsigned, lo) : 0u;
            pk[i][2 * j + 1] = ok ? __builtin_bit_cast(unsigned, hi) : 0u;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the residual reads
    __builtin_amdgcn_s_barrier();                        // every wave is done reading X
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int yr = (i == NF - 1 ? row_last : row_first + i * 16) + li;
        const unsigned ya = lds0 + 2 * BBF_W + yr * ROWB + g * 24;
#pragma unroll
        for (int v = 0; v < 3; ++v)
            asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(ya), "v"(u32x2{pk[i][2 * v], pk[i][2 * v + 1]}), "i"(v * 8) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // Y is complete
}

__device__ __forceinline__ void bbf_run(const Conv3Problem &p, const int mt0, const int tiles_this_block, const int nb, char *smem) {
    constexpr int KS = 48, NRB = 3, ROWB = 96, NCH = 14, BM = 512, MR = 4, NT = 512;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int m = nb * p.hpwp;
    const int mtiles = (m + BM - 1) / BM;
    int ntile = mtiles - mt0;
    if (ntile > tiles_this_block) ntile = tiles_this_block;
    if (ntile <= 0) return;
    const int halo = p.wp + 1;
    const int xrows = BM + 4 * halo, yrows = BM + 2 * halo;
    const int xunits = xrows * 6;
    // conv1's pixel fragments: nfr of them, dealt to the waves base or base + 1 each, contiguous
    const int nfr = (yrows + 15) >> 4, base = nfr >> 3, extra = nfr & 7;
    const int cnt = base + (wave < extra ? 1 : 0);
    const int f0 = wave * base + (wave < extra ? wave : extra);
    const int row_first = f0 * 16;
    int row_last = (f0 + cnt - 1) * 16;
    if (row_last > yrows - 16) row_last = yrows - 16;   // the last fragment ends on the last row (recomputes a few)
    const gcu16 in = (gcu16)p.in;
    const gu16 out = (gu16)p.out;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    char *const xy = smem + 2 * BBF_W;

    int xoff[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        int k0 = 32 * c + 8 * g;
        if (k0 >= 9 * KS) k0 = 0;
        const int tap = k0 / KS, ci = k0 - tap * KS;
        const int dh = tap / 3, dw = tap - 3 * dh;
        xoff[c] = (dh * p.wp + dw) * ROWB + (ci >> 3) * 16;
    }
    const int ch0 = g * 4 * NRB;
    // the 48 biases of each convolution sit one per lane; a lane picks its twelve with ds_bpermute when it needs them
    // (24 registers less across the tile loop -- the next tile's X is parked in registers during conv2)
    const int bl = lane < KS ? lane : 0;
    const int bvec1 = __float_as_int(((const GLOBAL_AS float *)p.bias)[bl]);
    const int bvec2 = __float_as_int(((const GLOBAL_AS float *)p.bias2)[bl]);
    // X of tile tt -> XY.  Rows outside the tensor's guard bands are clamped to a mapped row: whatever they hold only
    // reaches Y rows outside [0, m), which are zeroed.
    auto load_x = [&](int tt) {
        const long row0 = (long)(mt0 + tt) * BM - 2 * halo;
        const long lo = -(long)halo, hi = (long)m + halo + 511;
#pragma unroll
        for (int k = 0; k < BBF_XY / 16 / NT + 1; ++k) {
            if (k * NT + wave * 64 < xunits) {
                int u = k * NT + tid;
                asm volatile("" : "+v"(u));  // recompute the address per tile: hoisted, the ten 64-bit offsets spill
                if (u >= xunits) u = xunits - 1;
                const int r = (int)(((unsigned)u * 43691u) >> 18);   // u / 6 for u < 2^16
                const int q8 = u - r * 6;
                long gr = row0 + r;
                gr = gr < lo ? lo : (gr > hi ? hi : gr);
                glds16((const GLOBAL_AS char *)(in + gr * KS + q8 * 8), xy + (k * NT + wave * 64) * 16);
            }
        }
    };
    // The same X, for the tiles after the first, through registers: requested when conv2 starts and written to XY
    // once conv2 has finished with Y -- conv2's compute time to land, nothing exposed but ten ds_write.
    // Pieces 0..7 (units < 4096) go through registers; pieces 8 and 9 land beyond Y's last row (660 rows = 3960 units
    // at most), which nothing reads during conv2: those go straight to their place by LDS-DMA.
    constexpr int NXP = BBF_XY / 16 / NT + 1, NXR = 8;
    u32x4 xpre[NXR];
    // unit u = k * 512 + tid of the X image is slot u % 6 of row u / 6; 512 = 85 * 6 + 2, so piece k follows from piece 0
    const int r0u = (int)(((unsigned)tid * 43691u) >> 18), q0u = tid - r0u * 6;
    auto x_src = [&](int tt, int k) {
        const int row0 = (mt0 + tt) * BM - 2 * halo;   // (all row numbers fit 32 bits: m < 2^27)
        int r = r0u, q8 = q0u;
        asm volatile("" : "+v"(r), "+v"(q8));          // derive per tile: hoisted, the ten row / slot pairs would spill
        r += 85 * k + (2 * k) / 6, q8 += (2 * k) % 6;
        if (q8 >= 6) q8 -= 6, ++r;
        if (r >= xrows) r = xrows - 1, q8 = 5;          // past the end: re-read the last unit
        int gr = row0 + r;
        const int lo = -halo, hi = m + halo + 511;
        gr = gr < lo ? lo : (gr > hi ? hi : gr);
        return in + ((long)gr * KS + q8 * 8);
    };
    auto fetch_x = [&](int tt) {
#pragma unroll
        // (unconditional: a piece past the end re-reads the last unit and is not written.  Plain loads, not inline asm:
        // should the register allocator ever spill one, the compiler waits for it first -- slower, never wrong)
        for (int k = 0; k < NXR; ++k) xpre[k] = *(const GLOBAL_AS u32x4 *)x_src(tt, k);
#pragma unroll
        for (int k = NXR; k < NXP; ++k)
            if (k * NT + wave * 64 < xunits) glds16((const GLOBAL_AS char *)x_src(tt, k), xy + (k * NT + wave * 64) * 16);
    };
    auto store_x = [&]() {
        const unsigned a0 = lds0 + 2 * BBF_W + tid * 16;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // registers and LDS-DMA alike
#pragma unroll
        for (int k = 0; k < NXR; ++k)
            if (k * NT + wave * 64 < xunits)
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a0), "v"(xpre[k]), "i"(k * NT * 16) : "memory");
    };
    {   // both weight images, once per block
        const GLOBAL_AS char *w1 = (const GLOBAL_AS char *)p.w, *w2 = (const GLOBAL_AS char *)p.w2;
#pragma unroll
        for (int k = 0; k < (BBF_W / 16 + NT - 1) / NT; ++k) {
            const int u0 = k * NT + wave * 64;
            if (u0 < BBF_W / 16) {
                glds16(w1 + (size_t)(u0 + lane) * 16, smem + u0 * 16);
                glds16(w2 + (size_t)(u0 + lane) * 16, smem + BBF_W + u0 * 16);
            }
        }
        load_x(0);
    }
#ifdef HRN_C3_TIMING
    long long t_w = 0, t_c1 = 0, t_c1p = 0, t_c2 = 0, t_post = 0, t_epi = 0;
    C3_T(t_begin);
#endif
    for (int tt = 0; tt < ntile; ++tt) {
        C3_T(tA);
        if (tt == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the block's LDS-DMA: weights and the first X
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // later tiles: this wave's share of X is written
        __builtin_amdgcn_s_barrier();  // X (and the weights) are in LDS for every wave
        const long p0 = (long)(mt0 + tt) * BM;
        C3_T(tB);
        long long tC = 0;
        u32x2 rpre[MR][3];
        const unsigned res_a = lds0 + 2 * BBF_W + (wave * 16 * MR + li + 2 * halo) * ROWB + g * 24;
        if (cnt == 6)
            bbf_conv1<6>(p, xoff, bvec1, lds0, row_first, row_last, p0 - halo, m, lane, res_a, rpre, tC);
        else if (cnt == 5)
            bbf_conv1<5>(p, xoff, bvec1, lds0, row_first, row_last, p0 - halo, m, lane, res_a, rpre, tC);
        else
            bbf_conv1<4>(p, xoff, bvec1, lds0, row_first, row_last, p0 - halo, m, lane, res_a, rpre, tC);
        C3_T(tD);
        // ---- conv2 over Y: the chunk loop of conv3_run with both weight parts resident
        if (tt + 1 < ntile) fetch_x(tt + 1);   // the next tile's X lands in registers meanwhile
        unsigned okbits2 = 0;                  // the epilogue's pad mask, ahead of the loop for the same reason as conv1's
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            const int q = (int)p0 + wave * 16 * MR + i * 16 + li;
            const int n_img = (int)(((unsigned long long)(unsigned)q * p.magic_hpwp) >> p.shift_hpwp);
            const int rem = q - n_img * p.hpwp;
            const int ho = (int)(((unsigned long long)(unsigned)rem * p.magic_wp) >> p.shift_wp);
            const int wo = rem - ho * p.wp;
            okbits2 |= ((q < m) && (ho < p.h) && (wo < p.wd)) ? 1u << i : 0u;
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[MR][NRB];
        {
            float bias2[4 * NRB];
#pragma unroll
            for (int c = 0; c < 4 * NRB; ++c) bias2[c] = __int_as_float(__builtin_amdgcn_ds_bpermute((ch0 + c) * 4, bvec2));
#pragma unroll
            for (int i = 0; i < MR; ++i)
#pragma unroll
                for (int j = 0; j < NRB; ++j) acc[i][j] = f32x4{bias2[j * 4], bias2[j * 4 + 1], bias2[j * 4 + 2], bias2[j * 4 + 3]};
        }
        {
            s16x8 wf[2][NRB], xf[2][MR];
            const unsigned wl_a = lds0 + BBF_W + lane * 16;
            unsigned sl_a = lds0 + 2 * BBF_W + (wave * 16 * MR + li) * ROWB;
            asm volatile("" : "+v"(sl_a));   // (as in conv1: keep the per-chunk addresses out of the loop-carried registers)
#define BBF_READ2(SET, C)                                                                                         \
    {                                                                                                             \
        _Pragma("unroll") for (int j = 0; j < NRB; ++j)                                                           \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[SET][j]) : "v"(wl_a), "i"(((C)*NRB + j) * 1024)); \
        const unsigned xa = sl_a + xoff[C];                                                                        \
        _Pragma("unroll") for (int i = 0; i < MR; ++i)                                                            \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[SET][i]) : "v"(xa), "i"(i * 16 * ROWB));       \
    }
            BBF_READ2(0, 0)
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int cur = c & 1, nxt = cur ^ 1;
                if (c + 1 < NCH) {
                    BBF_READ2(nxt, c + 1)
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(NRB + MR) : "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MR; ++i)
#pragma unroll
                    for (int j = 0; j < NRB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[cur][j]),
                                                                            __builtin_bit_cast(bf16x8, xf[cur][i]), acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#undef BBF_READ2
        }
        C3_T(tE);
        __builtin_amdgcn_s_barrier();  // every wave is done reading Y: XY may be refilled
        if (tt + 1 < ntile) store_x();
        __builtin_amdgcn_sched_barrier(0);
        C3_T(tF);
        // ---- epilogue of conv2 (as in conv3_run): + residual, ReLU, zero on pad pixels, 24 contiguous bytes per lane
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            const int q = (int)p0 + wave * 16 * MR + i * 16 + li;
            const bool ok = (okbits2 >> i) & 1u;
            const size_t o = (size_t)q * KS + ch0;
            unsigned pk[2 * NRB];
#pragma unroll
            for (int j = 0; j < NRB; ++j) {
                const unsigned r01 = rpre[i][j][0], r23 = rpre[i][j][1];
                float v0 = acc[i][j][0] + __uint_as_float(r01 << 16);
                float v1 = acc[i][j][1] + __uint_as_float(r01 & 0xffff0000u);
                float v2 = acc[i][j][2] + __uint_as_float(r23 << 16);
                float v3 = acc[i][j][3] + __uint_as_float(r23 & 0xffff0000u);
                if (p.relu) v0 = relu1(v0), v1 = relu1(v1), v2 = relu1(v2), v3 = relu1(v3);
                typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                const bf16x2 lo = {(__bf16)v0, (__bf16)v1}, hi = {(__bf16)v2, (__bf16)v3};
                pk[2 * j] = ok ? __builtin_bit_cast(unsigned, lo) : 0u;
                pk[2 * j + 1] = ok ? __builtin_bit_cast(unsigned, hi) : 0u;
            }
            *(GLOBAL_AS u32x4 *)(out + o) = u32x4{pk[0], pk[1], pk[2], pk[3]};
            *(GLOBAL_AS u32x2 *)(out + o + 8) = u32x2{pk[4], pk[5]};
        }
#ifdef HRN_C3_TIMING
        C3_T(tG);
        t_w += tB - tA, t_c1 += tC - tB, t_c1p += tD - tC, t_c2 += tE - tD, t_post += tF - tE, t_epi += tG - tF;
#endif
    }
#ifdef HRN_C3_TIMING
    if (lane == 0 && g_c3_timing) {
        C3_T(t_end);
        long long *o = g_c3_timing + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[0] = t_w, o[1] = t_c1, o[2] = t_c1p, o[3] = t_c2, o[4] = t_end - t_begin, o[5] = ntile, o[6] = 100 + cnt,
        o[7] = t_post | (t_epi << 32);
    }
#endif
}

#include "conv3x3_n96.inc"

template <int KS, int NRB>
__global__ __launch_bounds__(512, 2) void conv3x3_lds_kernel(const Conv3Problem *__restrict__ probs,
                                                             const int2 *__restrict__ blockmap, const int nb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using CFG = C3Cfg<KS, NRB>;
#ifdef HRN_Q_TIMING   // debug (tools/c3q_test.hip): which CU ran this block, from when to when (s_memrealtime, 100 MHz)
    struct Stamp {
        long long t0;
        __device__ Stamp() : t0((long long)__builtin_amdgcn_s_memrealtime()) {}
        __device__ ~Stamp() {
            if (g_q_timing && threadIdx.x == 0) {
                unsigned hw, xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
                const int seq = g_q_seq;
                long long *o = g_q_timing + ((size_t)(seq & (kQSlots - 1)) * kQBlocks + (blockIdx.x < kQBlocks ? blockIdx.x : kQBlocks - 1)) * 4;
                o[0] = t0, o[1] = (long long)gridDim.x | ((long long)seq << 32), o[2] = (long long)__builtin_amdgcn_s_memrealtime();
                o[3] = (long long)(((xcc & 15u) << 16) | (hw & 0xff00u));   // XCC, SE / SH / CU bits of HW_ID
            }
        }
    } stamp_;
#endif
    const int2 bm = blockmap[blockIdx.x];
    // block map entry: x = problem | cout tile << 8 | M tiles of this block << 16,  y = first M tile
    const Conv3Problem p = probs[bm.x & 0xff];
    const int nt = (bm.x >> 8) & 0xff, tiles = bm.x >> 16;
    // y = first M tile | small << 30.  small: 128-pixel tiles (MR = 1) -- the host asks for them when even one tile
    // per block would leave CUs idle (a few crops): four times the blocks, a quarter of the MFMAs on a block's serial path
    const int mt0 = bm.y & 0x1fffffff;
    if constexpr (KS == 48 && NRB == 3) {
        if (bm.y & (1 << 29)) {  // a fused BasicBlock (bbf_run): 512-pixel tiles, both convolutions
            bbf_run(p, mt0, tiles, nb, smem);
            return;
        }
        if (p.n96) {  // 96 couts per block, 32-channel slices (conv3x3_n96.inc)
            if (p.compact) {   // tiles of real pixels only (the host sets it for bm == 512 geometries with enough padding to pay)
                if (bm.y >> 30)
                    c3n_run<1, true>(p, nt, mt0, tiles, nb, smem);
                else
                    c3n_run<4, true>(p, nt, mt0, tiles, nb, smem);
                return;
            }
            if (bm.y >> 30)
                c3n_run<1>(p, nt, mt0, tiles, nb, smem);
            else if (p.bm == 512)
                c3n_run<4>(p, nt, mt0, tiles, nb, smem);
            else
                c3n_run<3>(p, nt, mt0, tiles, nb, smem);
            return;
        }
    }
    if (bm.y >> 30) {
        conv3_run<CFG, 1>(p, nt, mt0, tiles, nb, smem);
    } else if constexpr (NRB == 4) {  // 64 accumulator + 64 fragment registers at MR = 4 would spill: 384-pixel tiles only
        conv3_run<CFG, 3>(p, nt, mt0, tiles, nb, smem);
    } else {
        if (p.bm == 512)
            conv3_run<CFG, 4>(p, nt, mt0, tiles, nb, smem);
        else
            conv3_run<CFG, 3>(p, nt, mt0, tiles, nb, smem);  // bm == 384
    }
}

#ifdef HRN_C3_TIMING
// debug: per-block phase cycle counters of the most recent launch (tools/c3_timing.py)
extern "C" int hrn_debug_c3_timing(long long *host_out, int max_blocks) {
    static long long *buf = nullptr;
    if (!buf) {
        if (hipMalloc((void **)&buf, (size_t)max_blocks * 512) != hipSuccess) return -1;
        (void)hipMemset(buf, 0, (size_t)max_blocks * 512);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_c3_timing), &buf, sizeof(buf));
        return 0;
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(host_out, buf, (size_t)max_blocks * 512, hipMemcpyDeviceToHost);
    return 1;
}
#endif

#ifdef HRN_Q_TIMING
// first call: allocate + arm; later calls: copy out [slot][block][4] = {t0, grid | seq << 32, t1, CU id}
extern "C" int hrn_debug_q_timing(long long *host_out) {
    static long long *buf = nullptr;
    const size_t bytes = (size_t)kQSlots * kQBlocks * 4 * sizeof(long long);
    if (!buf) {
        if (hipMalloc((void **)&buf, bytes) != hipSuccess) return -1;
        (void)hipMemset(buf, 0, bytes);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_q_timing), &buf, sizeof(buf));
        return 0;
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(host_out, buf, bytes, hipMemcpyDeviceToHost);
    return 1;
}
#endif

template <int KS, int NRB>
static hipError_t launch_c3(const Conv3Problem *probs_dev, const int2 *blockmap_dev, int nblocks, int nb, hipStream_t s) {
    using CFG = C3Cfg<KS, NRB>;
    // the <48, 3> launches may carry fused BasicBlocks (bbf_run), which lay LDS out differently and use all of it
    constexpr int LDS = (KS == 48 && NRB == 3) ? (BBF_LDS > N96_LDS ? BBF_LDS : N96_LDS) : CFG::LDS;
    static_assert(LDS >= CFG::LDS, "LDS budget");
    static std::atomic<unsigned long long> lds_set{0};   // per device: kernels.h set_dynamic_lds
    {
        const hipError_t e = set_dynamic_lds((const void *)conv3x3_lds_kernel<KS, NRB>, LDS, lds_set);
        if (e != hipSuccess) return e;
    }
#ifdef HRN_Q_TIMING
    hipLaunchKernelGGL(q_seq_bump, dim3(1), dim3(1), 0, s);
#endif
    hipLaunchKernelGGL((conv3x3_lds_kernel<KS, NRB>), dim3(nblocks), dim3(512), LDS, s, probs_dev, blockmap_dev, nb);
    return hipGetLastError();
}

int conv3x3_n96_ch64() { return N96_CH64; }
int conv3x3_n96_max_rows() { return N96_MAXROWS; }

int conv3x3_lds_bbf_ok(int wp) { return (512 + 4 * (wp + 1)) * 6 <= BBF_XY / 16; }

// pixels per M tile for a (KS, wp) pair: 512, or 384 when two 512-row slabs (+ halo) would not fit in LDS; 0 = unsupported
int conv3x3_lds_bm(int ks, int nrb, int wp) {
    if (ks == 16) return conv3x3_f32_bm(wp);   // fp32 kernel
    // (96-cout form: the last LDS-DMA piece of a slab is written in whole 16-row chunks -- the rows, rounded up to 16, must fit the buffer)
    if (ks == 32 && nrb == 6) {
        auto fits = [&](int bm) { return (bm + 2 * wp + 2 + 15) / 16 * 16 <= N96_MAXROWS; };
        return fits(512) ? 512 : fits(384) ? 384 : 0;
    }
    const int maxrows = ks == 48 ? C3Cfg<48, 3>::MAXROWS : C3Cfg<32, 4>::MAXROWS;
    if (nrb != 4 && 512 + 2 * wp + 2 <= maxrows) return 512;
    if (384 + 2 * wp + 2 <= maxrows) return 384;
    return 0;
}

hipError_t launch_conv3x3_lds(const Conv3Problem *probs_dev, const void *blockmap_dev, int nblocks, int nb, int ks,
                              int nrb, hipStream_t s) {
    if (nblocks <= 0) return hipSuccess;
    const int2 *bm = (const int2 *)blockmap_dev;
    if (ks == 16) return launch_conv3x3_f32(probs_dev, blockmap_dev, nblocks, nb, nrb, s);
    if ((ks == 48 && nrb == 3) || (ks == 32 && nrb == 6)) return launch_c3<48, 3>(probs_dev, bm, nblocks, nb, s);
    if (ks == 32 && nrb == 4) return launch_c3<32, 4>(probs_dev, bm, nblocks, nb, s);
    if (ks == 32 && nrb == 3) return launch_c3<32, 3>(probs_dev, bm, nblocks, nb, s);
    if (ks == 32 && nrb == 2) return launch_c3<32, 2>(probs_dev, bm, nblocks, nb, s);
    return hipErrorInvalidValue;
}

}  // namespace hrn
