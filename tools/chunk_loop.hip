// Micro-benchmark of the BasicBlock kernel's inner loop in isolation: 8 waves per CU, operands resident in LDS,
// 7 ds_read_b128 + 12 MFMA (16x16x32 bf16) per chunk and wave, nothing else.  Variants of how the two waves of a
// SIMD are scheduled against each other.   hipcc --offload-arch=gfx950 -O3 -o chunk_loop chunk_loop.hip
//   V0  as shipped: per chunk  read(c+1) ; wait(c) ; 12 MFMA     barrier every 7 chunks
//   V1  V0 without the barrier
//   V2  two-chunk segments:    read(c+2, c+3) ; wait ; 24 MFMA   (no own overlap; the SIMD partner covers)
//   V3  ping-pong: waves 0-3 run 24 MFMA while waves 4-7 read their next two chunks, barrier, swap
//   V4  V0 with waves 4-7 started half a chunk late (s_sleep)
//   V5  V0 + a tile epilogue every TH half-stages: 8 residual loads requested a half-stage ahead, unpack/add/ReLU/pack,
//       8 stores (the S = TH/2 branch of the real kernel without its LDS-DMA)
//   V6  V5 + LDS-DMA: 12 pieces per wave in even half-stages, 3 in odd ones, two per chunk
//   V7  V0 + that LDS-DMA only
//   V10 V7's bytes through registers instead: global_load_dwordx4 now, ds_write_b128 three chunks later
//   V11 role split: waves 0-3 compute a 128-pixel tile each (MR = 8: 11 reads + 24 MFMA per chunk) and issue no vector
//       memory; waves 4-7 only issue the LDS-DMA (24 / 6 pieces each); V12 = V11 + the compute waves' epilogue
//   V8  V7 with the LDS-DMA issued by waves 0-3 only (twice as many each); V9: by waves 0-3 in even chunks, 4-7 in odd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int MR = 4, NRB = 3, CPP = 7, ROWB = 96;

#define RD_W(SET, C)                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < NRB; ++j)                                                                     \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[SET][j]) : "v"(wl_a), "i"((((C) % CPP) * NRB + j) * 1024));
#define RD_X(SET, C)                                                                                                   \
    {                                                                                                                  \
        const unsigned xa = sl_a + xoff[(C) % CPP];                                                                    \
        _Pragma("unroll") for (int i = 0; i < MR; ++i)                                                                 \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[SET][i]) : "v"(xa), "i"(i * 16 * ROWB));            \
    }
#define MMA(SET)                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < MR; ++i) _Pragma("unroll") for (int j = 0; j < NRB; ++j)                     \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[SET][j]),                    \
                                                            __builtin_bit_cast(bf16x8, xf[SET][i]), acc[i][j], 0, 0, 0);

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#define GAS __attribute__((address_space(1)))
__device__ __forceinline__ void glds16(const GAS char *src, char *dst) {
    __builtin_amdgcn_global_load_lds((const GAS void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
}

template <int V, int TH>
__global__ __launch_bounds__(512, 2) void loop_kernel(float *out, int halves, const char *gsrc, char *gdst, size_t gbytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    for (int i = tid; i < 150 * 1024 / 4; i += 512) ((unsigned *)smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned wl_a = lds0 + lane * 16;
    const unsigned sl_a = lds0 + 43008 + (wave * 64 + li) * ROWB;
    int xoff[CPP];
#pragma unroll
    for (int c = 0; c < CPP; ++c) {
        const int k0 = 32 * c + 8 * g, tap = k0 / 48, ci = k0 - tap * 48;
        xoff[c] = ((tap / 3) * 97 + tap % 3) * ROWB + (ci >> 3) * 16;
    }
    f32x4 acc[MR][NRB];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NRB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    s16x8 wf[2][NRB], xf[2][MR];
    if (V == 4 && wave >= 4) __builtin_amdgcn_s_sleep(2);
    u32x4 rp4[MR];
    u32x2 rp2[MR];
    u32x4 stg[6];
    const size_t span = gbytes / gridDim.x;   // this block's private slice of the big buffers
    const GAS char *bsrc = (const GAS char *)gsrc + (size_t)blockIdx.x * span;
    GAS char *bdst = (GAS char *)gdst + (size_t)blockIdx.x * span;
    size_t roff = 0, doff = 0, woff = 0;
    for (int h = 0; h < halves; ++h) {
        if (V >= 5) {
            __builtin_amdgcn_s_barrier();
            const bool last = (h % TH) == TH - 1;
            if (V < 7 && last) {  // residual request: lands under this half-stage's MFMAs
#pragma unroll
                for (int i = 0; i < MR; ++i) {
                    const GAS char *rp = bsrc + (roff + (size_t)(wave * 64 + i * 16 + li) * 96 + g * 24) % (span - 4096);
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rp4[i]) : "v"(rp));
                    asm volatile("global_load_dwordx2 %0, %1, off offset:16" : "=v"(rp2[i]) : "v"(rp));
                }
                roff += 512 * 96;
            }
            RD_W(0, 0) RD_X(0, 0)
#pragma unroll
            for (int c = 0; c < CPP; ++c) {
                if (c + 1 < CPP) { RD_W((c + 1) & 1, c + 1) RD_X((c + 1) & 1, c + 1) }
                if (V == 10) {
                    const int np = (h & 1) ? 3 : 12;
                    // write what was loaded three chunks ago (ring of 3 x 2 registers), then load this chunk's two
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int slot = (c % 3) * 2 + t;
                        if (c >= 3 && 2 * (c - 3) + t < np) {
                            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(5 - t) : "memory");
                            asm volatile("ds_write_b128 %0, %1" ::"v"(lds0 + 100 * 1024 + ((2 * (c - 3) + t) * 8 + wave) * 1024 % (48 * 1024) + lane * 16), "v"(stg[slot]) : "memory");
                        }
                        if (2 * c + t < np) {
                            const GAS char *sp = bsrc + (doff + (size_t)(wave * 64 + lane) * 16) % (span - 4096);
                            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stg[slot]) : "v"(sp));
                            doff += 8192;
                        }
                    }
                } else if (V == 8 || V == 9) {
                    const int np = (h & 1) ? 6 : 24;
                    const bool mine = V == 8 ? wave < 4 : ((c & 1) == 0) == (wave < 4);
                    const int per = V == 8 ? 4 : 7;  // V9: a wave issues in every other chunk: up to 7 there
                    const int first = V == 8 ? 4 * c : (c >> 1) * 7;
                    if (mine) {
#pragma unroll
                        for (int t = 0; t < per; ++t)
                            if (first + t < np) {
                                glds16(bsrc + (doff + (size_t)(wave * 64 + lane) * 16) % (span - 4096), smem + 100 * 1024 + ((first + t) * 8 + wave) * 1024 % (48 * 1024));
                                doff += 8192;
                            }
                    }
                } else if (V >= 6) {  // LDS-DMA pieces: 12 per wave in even half-stages (two per chunk), 3 in odd ones
                    const int np = (h & 1) ? 3 : 12;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        if (2 * c + t < np) {
                            glds16(bsrc + (doff + (size_t)(wave * 64 + lane) * 16) % (span - 4096), smem + 100 * 1024 + ((2 * c + t) * 8 + wave) * 1024 % (48 * 1024));
                            doff += 8192;
                        }
                }
                if (c + 1 < CPP) asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(NRB + MR) : "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                MMA(c & 1)
                __builtin_amdgcn_sched_barrier(0);
            }
            if (V < 7 && last) {  // epilogue
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < MR; ++i) {
                    unsigned pk[6];
#pragma unroll
                    for (int j = 0; j < NRB; ++j) {
                        const unsigned r01 = j < 2 ? rp4[i][2 * j] : rp2[i][0], r23 = j < 2 ? rp4[i][2 * j + 1] : rp2[i][1];
                        float v0 = fmaxf(acc[i][j][0] + __uint_as_float(r01 << 16), 0.f), v1 = fmaxf(acc[i][j][1] + __uint_as_float(r01 & 0xffff0000u), 0.f);
                        float v2 = fmaxf(acc[i][j][2] + __uint_as_float(r23 << 16), 0.f), v3 = fmaxf(acc[i][j][3] + __uint_as_float(r23 & 0xffff0000u), 0.f);
                        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                        const bf16x2 lo = {(__bf16)v0, (__bf16)v1}, hi = {(__bf16)v2, (__bf16)v3};
                        pk[2 * j] = __builtin_bit_cast(unsigned, lo), pk[2 * j + 1] = __builtin_bit_cast(unsigned, hi);
                        acc[i][j] = f32x4{0.5f, 0.25f, 0.125f, 1.f};
                    }
                    GAS char *o = bdst + (woff + (size_t)(wave * 64 + i * 16 + li) * 96 + g * 24) % (span - 4096);
                    *(GAS u32x4 *)o = u32x4{pk[0], pk[1], pk[2], pk[3]};
                    *(GAS u32x2 *)(o + 16) = u32x2{pk[4], pk[5]};
                }
                woff += 512 * 96;
            }
        } else if (V == 0 || V == 4) {
            __builtin_amdgcn_s_barrier();
            RD_W(0, 0) RD_X(0, 0)
#pragma unroll
            for (int c = 0; c < CPP; ++c) {
                if (c + 1 < CPP) { RD_W((c + 1) & 1, c + 1) RD_X((c + 1) & 1, c + 1) }
                if (c + 1 < CPP) asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(NRB + MR) : "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                MMA(c & 1)
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (V == 1) {
            RD_W(0, 0) RD_X(0, 0)
#pragma unroll
            for (int c = 0; c < CPP; ++c) {
                if (c + 1 < CPP) { RD_W((c + 1) & 1, c + 1) RD_X((c + 1) & 1, c + 1) }
                if (c + 1 < CPP) asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(NRB + MR) : "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                MMA(c & 1)
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (V == 2) {  // 8 chunks per iteration in segments of two
#pragma unroll
            for (int c = 0; c < 8; c += 2) {
                RD_W(0, c) RD_X(0, c) RD_W(1, c + 1) RD_X(1, c + 1)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                MMA(0) MMA(1)
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (V == 3) {  // ping-pong over 8 chunks: even phases waves 0-3 compute / 4-7 load, odd phases swapped
            if (h == 0 && wave < 4) {
                RD_W(0, 0) RD_X(0, 0) RD_W(1, 1) RD_X(1, 1)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int ph = 0; ph < 8; ++ph) {
                __builtin_amdgcn_s_barrier();
                const bool compute = ((ph & 1) == 0) == (wave < 4);
                if (compute) {
                    MMA(0) MMA(1)
                } else {
                    RD_W(0, ph) RD_X(0, ph) RD_W(1, ph + 1) RD_X(1, ph + 1)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NRB; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) out[tid] = s;
}

template <int V, int TH>
__global__ __launch_bounds__(512, 2) void split_kernel(float *out, int halves, const char *gsrc, char *gdst, size_t gbytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MR8 = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    for (int i = tid; i < 150 * 1024 / 4; i += 512) ((unsigned *)smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const size_t span = gbytes / gridDim.x;
    const GAS char *bsrc = (const GAS char *)gsrc + (size_t)blockIdx.x * span;
    GAS char *bdst = (GAS char *)gdst + (size_t)blockIdx.x * span;
    if (wave >= 4) {  // loader waves
        size_t doff = 0;
        for (int h = 0; h < halves; ++h) {
            __builtin_amdgcn_s_barrier();
            const int np = (h & 1) ? 6 : 24;
            for (int t = 0; t < np; ++t) {
                glds16(bsrc + (doff + (size_t)((wave - 4) * 64 + lane) * 16) % (span - 4096), smem + 100 * 1024 + (t * 4 + (wave - 4)) * 1024 % (48 * 1024));
                doff += 4096;
                if ((t & 3) == 3) __builtin_amdgcn_s_sleep(4);  // spread over the half-stage
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        return;
    }
    const unsigned wl_a = lds0 + lane * 16;
    const unsigned sl_a = lds0 + 43008 + (wave * 128 + li) * ROWB;
    int xoff[CPP];
#pragma unroll
    for (int c = 0; c < CPP; ++c) {
        const int k0 = 32 * c + 8 * g, tap = k0 / 48, ci = k0 - tap * 48;
        xoff[c] = ((tap / 3) * 97 + tap % 3) * ROWB + (ci >> 3) * 16;
    }
    f32x4 acc[MR8][NRB];
#pragma unroll
    for (int i = 0; i < MR8; ++i)
#pragma unroll
        for (int j = 0; j < NRB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    s16x8 wf[2][NRB], xf[2][MR8];
    u32x4 rp4[MR8];
    u32x2 rp2[MR8];
    size_t roff = 0, woff = 0;
#define RD_X8(SET, C)                                                                                                  \
    {                                                                                                                  \
        const unsigned xa = sl_a + xoff[(C) % CPP];                                                                    \
        _Pragma("unroll") for (int i = 0; i < MR8; ++i)                                                                \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[SET][i]) : "v"(xa), "i"(i * 16 * ROWB));            \
    }
    for (int h = 0; h < halves; ++h) {
        __builtin_amdgcn_s_barrier();
        const bool last = (h % TH) == TH - 1;
        if (V == 12 && last) {
#pragma unroll
            for (int i = 0; i < MR8; ++i) {
                const GAS char *rp = bsrc + (roff + (size_t)(wave * 128 + i * 16 + li) * 96 + g * 24) % (span - 4096);
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rp4[i]) : "v"(rp));
                asm volatile("global_load_dwordx2 %0, %1, off offset:16" : "=v"(rp2[i]) : "v"(rp));
            }
            roff += 512 * 96;
        }
        RD_W(0, 0) RD_X8(0, 0)
#pragma unroll
        for (int c = 0; c < CPP; ++c) {
            if (c + 1 < CPP) { RD_W((c + 1) & 1, c + 1) RD_X8((c + 1) & 1, c + 1) }
            if (c + 1 < CPP) asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(NRB + MR8) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MR8; ++i)
#pragma unroll
                for (int j = 0; j < NRB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[c & 1][j]),
                                                                        __builtin_bit_cast(bf16x8, xf[c & 1][i]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (V == 12 && last) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < MR8; ++i) {
                unsigned pk[6];
#pragma unroll
                for (int j = 0; j < NRB; ++j) {
                    const unsigned r01 = j < 2 ? rp4[i][2 * j] : rp2[i][0], r23 = j < 2 ? rp4[i][2 * j + 1] : rp2[i][1];
                    float v0 = fmaxf(acc[i][j][0] + __uint_as_float(r01 << 16), 0.f), v1 = fmaxf(acc[i][j][1] + __uint_as_float(r01 & 0xffff0000u), 0.f);
                    float v2 = fmaxf(acc[i][j][2] + __uint_as_float(r23 << 16), 0.f), v3 = fmaxf(acc[i][j][3] + __uint_as_float(r23 & 0xffff0000u), 0.f);
                    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                    const bf16x2 lo = {(__bf16)v0, (__bf16)v1}, hi = {(__bf16)v2, (__bf16)v3};
                    pk[2 * j] = __builtin_bit_cast(unsigned, lo), pk[2 * j + 1] = __builtin_bit_cast(unsigned, hi);
                    acc[i][j] = f32x4{0.5f, 0.25f, 0.125f, 1.f};
                }
                GAS char *o = bdst + (woff + (size_t)(wave * 128 + i * 16 + li) * 96 + g * 24) % (span - 4096);
                *(GAS u32x4 *)o = u32x4{pk[0], pk[1], pk[2], pk[3]};
                *(GAS u32x2 *)(o + 16) = u32x2{pk[4], pk[5]};
            }
            woff += 512 * 96;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MR8; ++i)
#pragma unroll
        for (int j = 0; j < NRB; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) out[tid] = s;
}

static char *g_src = nullptr, *g_dst = nullptr;
static char *g_src_fwd, *g_dst_fwd;
static size_t g_bytes = (size_t)2 << 30;  // footprint of the global traffic (2 GiB: HBM; 64 MiB: stays in L2 + Infinity Cache)
template <int V, int TH>
static void run_split(const char *name) {
    float *d; hipMalloc(&d, 4096);
    if (!g_src) { hipMalloc(&g_src, g_bytes); hipMalloc(&g_dst, g_bytes); hipMemset(g_src, 0, g_bytes); }
    hipFuncSetAttribute((const void *)split_kernel<V, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    const int halves = 2000, blocks = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    split_kernel<V, TH><<<blocks, 512, 150 * 1024>>>(d, 10, g_src, g_dst, g_bytes);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    split_kernel<V, TH><<<blocks, 512, 150 * 1024>>>(d, halves, g_src, g_dst, g_bytes);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)blocks * 4 * halves * 7 * 24.0;
    const double tf = mfma * 16384.0 / (ms * 1e-3) / 1e12;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s issued  = %4.1f %% of 2.5 PF\n", name, ms, tf, 100 * tf / 2500);
    hipFree(d);
}

template <int V, int TH = 2>
static void run(const char *name, int chunks_per_half, int mfma_share_num, int mfma_share_den) {
    float *d; hipMalloc(&d, 4096);
    if (!g_src) { hipMalloc(&g_src, g_bytes); hipMalloc(&g_dst, g_bytes); hipMemset(g_src, 0, g_bytes); }
    hipFuncSetAttribute((const void *)loop_kernel<V, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    const int halves = 2000, blocks = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    loop_kernel<V, TH><<<blocks, 512, 150 * 1024>>>(d, 10, g_src, g_dst, g_bytes);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    loop_kernel<V, TH><<<blocks, 512, 150 * 1024>>>(d, halves, g_src, g_dst, g_bytes);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // MFMAs per wave per half: chunks * 12, times the share of waves computing (ping-pong: every wave computes half the phases)
    const double mfma = (double)blocks * 8 * halves * chunks_per_half * 12.0 * mfma_share_num / mfma_share_den;
    const double tf = mfma * 16384.0 / (ms * 1e-3) / 1e12;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s issued  = %4.1f %% of 2.5 PF\n", name, ms, tf, 100 * tf / 2500);
    hipFree(d);
}

int main() {
    run<0>("V0 shipped: read(c+1);wait;12 MFMA +barrier/7", 7, 1, 1);
    run<1>("V1 same, no barrier", 7, 1, 1);
    run<2>("V2 two-chunk segments (24 MFMA per wait)", 8, 1, 1);
    run<3>("V3 ping-pong halves with barriers", 8, 1, 2);
    run<4>("V4 shipped, waves 4-7 start late", 7, 1, 1);
    run<0>("V0 again (first run pays the clock ramp)", 7, 1, 1);
    run<1>("V1 again", 7, 1, 1);
    run<5, 2>("V5 + epilogue every 2 halves (S=1)", 7, 1, 1);
    run<5, 4>("V5 + epilogue every 4 halves (S=2)", 7, 1, 1);
    run<5, 16>("V5 + epilogue every 16 halves (S=8)", 7, 1, 1);
    run<7, 2>("V7 + LDS-DMA only (12 / 3 pieces per wave)", 7, 1, 1);
    run<6, 2>("V6 + epilogue/2 + LDS-DMA", 7, 1, 1);
    run<6, 4>("V6 + epilogue/4 + LDS-DMA", 7, 1, 1);
    run<6, 16>("V6 + epilogue/16 + LDS-DMA", 7, 1, 1);
    g_bytes = (size_t)64 << 20;   // 64 KiB per block: every global access hits L2
    printf("-- same with a 64 MiB footprint (L2-resident) --\n");
    run<5, 2>("V5 + epilogue every 2 halves (S=1)", 7, 1, 1);
    run<5, 4>("V5 + epilogue every 4 halves (S=2)", 7, 1, 1);
    run<7, 2>("V7 + LDS-DMA only (12 / 3 pieces per wave)", 7, 1, 1);
    run<10, 2>("V10 same bytes via registers + ds_write", 7, 1, 1);
    run<8, 2>("V8 LDS-DMA by waves 0-3 only (24 / 6)", 7, 1, 1);
    run<9, 2>("V9 LDS-DMA alternating halves per chunk", 7, 1, 1);
    run<6, 4>("V6 + epilogue/4 + LDS-DMA", 7, 1, 1);
    run_split<11, 2>("V11 role split, LDS-DMA by loader waves");
    run_split<12, 2>("V12 role split + epilogue/2 in compute waves");
    run_split<12, 4>("V12 role split + epilogue/4");
    return 0;
}
