// PROTOTYPE harness: tools/c3n_test.hip for the half-block (4 waves, 256 pixels, two blocks per CU) form in conv3x3_n96h.inc.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o tools/bin/c3h_test tools/proto/c3h_test.hip
//   c3h_test [crops]   (every output element against a naive convolution, then timing; the shipped form timed beside it)
#include "../../simple-hrnet_amd/csrc/conv3x3_lds.hip"
namespace hrn {
int conv3x3_f32_bm(int) { return 0; }
hipError_t launch_conv3x3_f32(const Conv3Problem *, const void *, int, int, int, hipStream_t) { return hipErrorInvalidValue; }
}  // namespace hrn
namespace hrn {
#include "conv3x3_n96h.inc"
#include "conv3x3_n96l.inc"
}
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
using namespace hrn;

static inline uint16_t f2bf_h(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf2f_h(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline void fast_div_h(int d, unsigned *magic, int *shift) {
    int l = 0;
    while ((1 << l) < d) ++l;
    *shift = 30 + l;
    *magic = (unsigned)((1ull << *shift) / (unsigned)d + 1);
}

// reference: one thread per (row, cout); K order irrelevant (fp32 accumulate, compared with a tolerance)
__global__ void ref_conv(const uint16_t *in, const uint16_t *w /*[cout][9][cin] bf16*/, const float *bias, const uint16_t *res,
                         uint16_t *out, int m, int cin, int cout, int h, int wd, int wp, int hpwp, int relu) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)m * cout) return;
    const int q = (int)(idx / cout), co = (int)(idx % cout);
    const int rem = q % hpwp, ho = rem / wp, wo = rem % wp;
    float acc = bias[co];
    if (ho < h && wo < wd) {
        for (int t = 0; t < 9; ++t) {
            const long r = (long)q + (t / 3 - 1) * wp + (t % 3 - 1);
            const uint16_t *x = in + r * cin;
            const uint16_t *ww = w + ((size_t)co * 9 + t) * cin;
            for (int c = 0; c < cin; ++c) acc += __uint_as_float((unsigned)x[c] << 16) * __uint_as_float((unsigned)ww[c] << 16);
        }
        if (res) acc += __uint_as_float((unsigned)res[(size_t)q * cout + co] << 16);
        if (relu) acc = fmaxf(acc, 0.f);
    } else {
        acc = 0.f;
    }
    unsigned u = __float_as_uint(acc);
    u += 0x7fffu + ((u >> 16) & 1u);
    out[(size_t)q * cout + co] = (uint16_t)(u >> 16);
}

static int g_skew = 0, g_lds = N96H_LDS;   // (g_lds = 96 KiB: one block per CU)
static hipError_t launch_h(const Conv3Problem *dp, const int2 *dmap, int blocks, int nb) {
    static bool once = false;
    if (!once) {
        hipFuncSetAttribute((const void *)conv3x3_n96h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        once = true;
    }
    conv3x3_n96h_kernel<<<blocks, 256, g_lds, 0>>>(dp, dmap, nb, g_skew);
    return hipGetLastError();
}
static hipError_t launch_w(const Conv3Problem *dp, const int2 *dmap, int blocks, int nb) {
    static bool once = false;
    if (!once) {
        hipFuncSetAttribute((const void *)conv3x3_n96w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, N96W_LDS);
        once = true;
    }
    conv3x3_n96w_kernel<<<blocks, 256, N96W_LDS, 0>>>(dp, dmap, nb);
    return hipGetLastError();
}
static int g_lmr = 6;   // loader-wave form: 6 = 384-pixel tiles, weights 3 stages ahead; 5 = 320-pixel tiles, 5 stages ahead
static hipError_t launch_l(const Conv3Problem *dp, const int2 *dmap, int blocks, int nb) {
    static bool once = false;
    if (!once) {
        hipFuncSetAttribute((const void *)conv3x3_n96l_kernel<6, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, N96LCfg<6, 4>::LDS);
        hipFuncSetAttribute((const void *)conv3x3_n96l_kernel<5, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, N96LCfg<5, 6>::LDS);
        once = true;
    }
    if (g_lmr == 5)
        conv3x3_n96l_kernel<5, 6><<<blocks, 512, N96LCfg<5, 6>::LDS, 0>>>(dp, dmap, nb);
    else
        conv3x3_n96l_kernel<6, 4><<<blocks, 512, N96LCfg<6, 4>::LDS, 0>>>(dp, dmap, nb);
    return hipGetLastError();
}
static int g_show = 0;   // print that many mismatches per case
static int g_cold = 0;   // > 1: that many tensor sets in rotation (operands from HBM, not from the Infinity Cache)
static int g_wide = 0;   // 2: the loader-wave form (conv3x3_n96l.inc: 384-pixel tiles, 8 waves)   // run_shape(half = true) launches the wide form (512-pixel tiles, one wave per SIMD) instead

struct Shape {
    int c, h, w;
};

static int run_shape(const Shape &sh, int nb, bool with_res, int tpb, bool small, int reps, bool half = true) {
    const int C = sh.c, H = sh.h, W = sh.w, wp = W + 1, hp = H + 1, hpwp = hp * wp;
    const int m = nb * hpwp;
    const int guard_front = wp + 1, guard_back = wp + 1 + 512;
    const size_t rows = (size_t)guard_front + m + guard_back;
    std::vector<uint16_t> hin(rows * C, 0), hres(rows * C, 0);
    srand(1234 + C);
    for (int n = 0; n < nb; ++n)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                for (int c = 0; c < C; ++c) {
                    const size_t r = (size_t)guard_front + (size_t)n * hpwp + y * wp + x;
                    hin[r * C + c] = f2bf_h((rand() % 2001 - 1000) / 1000.f);
                    hres[r * C + c] = f2bf_h((rand() % 2001 - 1000) / 500.f);
                }
    const int K = 9 * C;
    std::vector<float> wf((size_t)C * K);
    std::vector<uint16_t> wref((size_t)C * K);
    for (size_t i = 0; i < wf.size(); ++i) {
        wf[i] = (rand() % 2001 - 1000) / 1000.f / sqrtf((float)K) * 2.f;
        wref[i] = f2bf_h(wf[i]);   // [co][tap][ci]
    }
    std::vector<float> hb(C);
    for (int c = 0; c < C; ++c) hb[c] = (rand() % 2001 - 1000) / 2000.f;
    // pack as hrnet_mi355.cpp: pack_conv_lds with KS = 32, NRB = 6
    const int KS = 32, NRB = 6, slices = C / KS, ntiles = C / 96, nch = 9;
    std::vector<uint16_t> wpk((size_t)ntiles * slices * nch * NRB * 512);
    for (int t = 0; t < ntiles; ++t)
        for (int s = 0; s < slices; ++s) {
            uint16_t *blk = wpk.data() + ((size_t)t * slices + s) * nch * NRB * 512;
            for (int c = 0; c < nch; ++c)
                for (int j = 0; j < NRB; ++j)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int li = lane & 15, g = lane >> 4;
                        const int co = conv3x3_n96_ch64() ? t * 96 + (j >> 1) * 32 + (li >> 2) * 8 + (j & 1) * 4 + (li & 3)
                                                          : t * 16 * NRB + (li >> 2) * 4 * NRB + j * 4 + (li & 3);
                        uint16_t *d = blk + ((size_t)(c * NRB + j) * 64 + lane) * 8;
                        for (int e = 0; e < 8; ++e) {
                            const int kl = 32 * c + 8 * g + e;
                            const int tap = kl / KS, cil = kl % KS;
                            d[e] = f2bf_h(wf[(size_t)co * K + tap * C + s * KS + cil]);
                        }
                    }
        }
    uint16_t *din, *dres, *dout, *dref, *dw, *dwref;
    float *dbias;
    hipMalloc(&din, rows * C * 2), hipMalloc(&dres, rows * C * 2), hipMalloc(&dout, rows * C * 2), hipMalloc(&dref, rows * C * 2);
    hipMalloc(&dw, wpk.size() * 2), hipMalloc(&dwref, wref.size() * 2), hipMalloc(&dbias, C * 4);
    hipMemcpy(din, hin.data(), rows * C * 2, hipMemcpyHostToDevice);
    hipMemcpy(dres, hres.data(), rows * C * 2, hipMemcpyHostToDevice);
    hipMemset(dout, 0x7f, rows * C * 2), hipMemset(dref, 0, rows * C * 2);
    hipMemcpy(dw, wpk.data(), wpk.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dwref, wref.data(), wref.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dbias, hb.data(), C * 4, hipMemcpyHostToDevice);
    Conv3Problem p;
    memset(&p, 0, sizeof p);
    p.in = din + (size_t)guard_front * C, p.out = dout + (size_t)guard_front * C, p.w = dw, p.bias = dbias;
    p.res = with_res ? dres + (size_t)guard_front * C : nullptr;
    p.cin = C, p.cout = C, p.h = H, p.wd = W, p.wp = wp, p.hpwp = hpwp, p.relu = 1, p.slices = slices, p.ntiles = ntiles;
    p.tiles_per_block = tpb, p.bm = conv3x3_lds_bm(32, 6, wp);
    fast_div_h(hpwp, &p.magic_hpwp, &p.shift_hpwp), fast_div_h(wp, &p.magic_wp, &p.shift_wp);
    p.n96 = 1;
    if (p.bm == 0) {
        printf("shape unsupported\n");
        return 1;
    }
    const int bm = half ? (g_wide == 2 ? 64 * g_lmr : g_wide ? 512 : 256) : small ? 128 : p.bm;
    const int mtiles = (m + bm - 1) / bm;
    std::vector<int2> map;
    const int mgroups = (mtiles + tpb - 1) / tpb;
    const int RW = half && !g_wide ? 16 : 8;   // blocks of one M group range per round and cout tile
    for (int round = 0; round * RW < mgroups; ++round)
        for (int nt = 0; nt < ntiles; ++nt)
            for (int x = 0; x < RW; ++x) {
                const int mg = round * RW + x;
                if (mg >= mgroups) continue;
                const int tiles = std::min(tpb, mtiles - mg * tpb);
                map.push_back(int2{0 | (nt << 8) | (tiles << 16), (mg * tpb) | (small ? 1 << 30 : 0)});
            }
    Conv3Problem *dp;
    int2 *dmap;
    hipMalloc(&dp, sizeof p), hipMalloc(&dmap, map.size() * sizeof(int2));
    hipMemcpy(dp, &p, sizeof p, hipMemcpyHostToDevice);
    hipMemcpy(dmap, map.data(), map.size() * sizeof(int2), hipMemcpyHostToDevice);
    hipError_t e = half ? (g_wide == 2 ? launch_l(dp, dmap, (int)map.size(), nb) : g_wide ? launch_w(dp, dmap, (int)map.size(), nb) : launch_h(dp, dmap, (int)map.size(), nb)) : launch_conv3x3_lds(dp, dmap, (int)map.size(), nb, 32, 6, 0);
    hipError_t e2 = hipDeviceSynchronize();
    if (e != hipSuccess || e2 != hipSuccess) {
        printf("launch failed: %s / %s\n", hipGetErrorString(e), hipGetErrorString(e2));
        return 1;
    }
    const long total = (long)m * C;
    ref_conv<<<(unsigned)((total + 255) / 256), 256>>>(din + (size_t)guard_front * C, dwref, dbias, with_res ? dres + (size_t)guard_front * C : nullptr,
                                                       dref + (size_t)guard_front * C, m, C, C, H, W, wp, hpwp, 1);
    hipDeviceSynchronize();
    std::vector<uint16_t> ho(rows * C), hr(rows * C);
    hipMemcpy(ho.data(), dout, rows * C * 2, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), dref, rows * C * 2, hipMemcpyDeviceToHost);
    long bad = 0, first_bad = -1;
    double maxerr = 0;
    for (long q = 0; q < m; ++q)
        for (int c = 0; c < C; ++c) {
            const size_t i = ((size_t)guard_front + q) * C + c;
            const float a = bf2f_h(ho[i]), b = bf2f_h(hr[i]);
            const float err = fabsf(a - b);
            if (!(err <= 0.02f + 0.01f * fabsf(b))) {
                if (first_bad < 0) first_bad = q * C + c;
                if (bad < g_show) printf("   bad: row %ld ch %d got %.4f (0x%04x) want %.4f\n", q, c, a, ho[i], b);
                ++bad;
            }
            if (err > maxerr) maxerr = err;
        }
    // rows past m (tail guard) must have been written as zeros or left alone (0x7f7f pattern) -- never garbage
    long tail_bad = 0;
    for (size_t i = ((size_t)guard_front + m) * C; i < rows * C; ++i)
        if (ho[i] != 0 && ho[i] != 0x7f7f) ++tail_bad;
    for (size_t i = 0; i < (size_t)guard_front * C; ++i)
        if (ho[i] != 0x7f7f) ++tail_bad;
    // timing.  g_cold > 1: rotate over g_cold sets of (input, residual, output) tensors so that no launch finds its operands in the
    // 256-MB Infinity Cache -- the condition a convolution meets inside the net, where a layer's tensors total 0.3 GB
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float ms = 0;
    if (reps > 0) {
        auto launch = [&](const Conv3Problem *dpp) {
            return half ? (g_wide == 2 ? launch_l(dpp, dmap, (int)map.size(), nb) : g_wide ? launch_w(dpp, dmap, (int)map.size(), nb) : launch_h(dpp, dmap, (int)map.size(), nb))
                        : launch_conv3x3_lds(dpp, dmap, (int)map.size(), nb, 32, 6, 0);
        };
        const int nset = g_cold > 1 ? g_cold : 1;
        std::vector<Conv3Problem> hp(nset, p);
        std::vector<uint16_t *> extra;
        for (int k = 1; k < nset; ++k) {
            uint16_t *a, *b, *c;
            hipMalloc(&a, rows * C * 2), hipMalloc(&b, rows * C * 2), hipMalloc(&c, rows * C * 2);
            hipMemcpy(a, din, rows * C * 2, hipMemcpyDeviceToDevice), hipMemcpy(b, dres, rows * C * 2, hipMemcpyDeviceToDevice);
            hipMemset(c, 0, rows * C * 2);
            hp[k].in = a + (size_t)guard_front * C, hp[k].out = c + (size_t)guard_front * C;
            hp[k].res = with_res ? b + (size_t)guard_front * C : nullptr;
            extra.push_back(a), extra.push_back(b), extra.push_back(c);
        }
        Conv3Problem *dps;
        hipMalloc(&dps, nset * sizeof(Conv3Problem));
        hipMemcpy(dps, hp.data(), nset * sizeof(Conv3Problem), hipMemcpyHostToDevice);
        for (int i = 0; i < 3; ++i) launch(dps + i % nset);
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) launch(dps + i % nset);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        ms /= reps;
        for (auto q : extra) hipFree(q);
        hipFree(dps);
    }
    const double gflop = 2.0 * 9 * C * (double)C * H * W * nb * 1e-9;
    printf("%s C=%3d %3dx%-3d nb=%3d res=%d tpb=%d bm=%d blocks=%5zu : bad=%ld (first %ld) tail_bad=%ld maxerr=%.4f", half ? (g_wide == 2 ? "load" : g_wide ? "wide" : "half") : "full", C, H, W, nb, (int)with_res, tpb,
           bm, map.size(), bad, first_bad, tail_bad, maxerr);
    if (reps > 0) {
        const int rounds = ((int)map.size() + (half && !g_wide ? 511 : 255)) / (half && !g_wide ? 512 : 256);
        printf("  %.1f us  %.0f TFLOP/s (algorithmic)  [%d round(s) x %d stages: %.2f us per stage]", ms * 1e3, gflop / ms, rounds,
               tpb * slices * 3, ms * 1e3 / (rounds * tpb * slices * 3));
    }
    printf("\n");
    if (bad && first_bad >= 0) {
        const long q = first_bad / C;
        printf("   first bad: row %ld (img %ld, y %ld, x %ld) ch %ld\n", q, q / hpwp, (q % hpwp) / wp, (q % hpwp) % wp, first_bad % C);
    }
    hipFree(din), hipFree(dres), hipFree(dout), hipFree(dref), hipFree(dw), hipFree(dwref), hipFree(dbias), hipFree(dp), hipFree(dmap);
    return bad || tail_bad ? 1 : 0;
}

int main(int argc, char **argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 256;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;   // 0: half blocks + the shipped form, 1: the wide form only, 2: the loader-wave form (+ shipped)
    int fails = 0;
    if (mode == 3) {   // the forms with their operands in the Infinity Cache (one tensor set) and from HBM (four sets in rotation)
        for (int cold : {0, 4}) {
            g_cold = cold;
            printf("-- %s\n", cold ? "operands from HBM (4 tensor sets in rotation)" : "operands cache-resident (one tensor set)");
            g_wide = 0;
            fails += run_shape({96, 48, 36}, nb, true, 4, false, 20, false);
            fails += run_shape({192, 24, 18}, nb, true, 2, false, 20, false);
            fails += run_shape({384, 12, 9}, 252, true, 1, false, 20, false);
            fails += run_shape({96, 48, 36}, nb, true, 8, false, 20, true);
            fails += run_shape({192, 24, 18}, nb, true, 4, false, 20, true);
            fails += run_shape({384, 12, 9}, 252, true, 2, false, 20, true);
            g_wide = 2, g_lmr = 6;
            fails += run_shape({96, 48, 36}, nb, true, 6, false, 20);
            fails += run_shape({192, 24, 18}, nb, true, 3, false, 20);
            fails += run_shape({384, 12, 9}, 252, true, 2, false, 20);
        }
        printf(fails ? "FAILED (%d)\n" : "all shapes OK\n", fails);
        return fails ? 1 : 0;
    }
    if (mode == 2) {
        g_wide = 2;
        const int small_only = argc > 3 ? atoi(argv[3]) : 0;
        g_show = argc > 4 ? atoi(argv[4]) : 0;
        fails += run_shape({96, 16, 12}, 2, true, 1, false, 0);
        fails += run_shape({96, 48, 36}, 3, true, 2, false, 0);
        fails += run_shape({96, 48, 36}, 3, false, 1, false, 0);
        fails += run_shape({192, 24, 18}, 5, true, 3, false, 0);
        fails += run_shape({384, 12, 9}, 7, true, 1, false, 0);
        g_lmr = 5;
        fails += run_shape({96, 16, 12}, 2, true, 1, false, 0);
        fails += run_shape({96, 48, 36}, 3, true, 2, false, 0);
        fails += run_shape({96, 48, 36}, 3, false, 1, false, 0);
        fails += run_shape({192, 24, 18}, 5, true, 3, false, 0);
        fails += run_shape({384, 12, 9}, 7, true, 1, false, 0);
        if (!small_only) {
            for (int mr : {6, 5}) {
                g_lmr = mr;
                printf("-- %d-pixel tiles\n", 64 * mr);
                fails += run_shape({96, 48, 36}, nb, true, mr == 6 ? 6 : 7, false, 20);    // one block per CU: 202 / 208 blocks
                fails += run_shape({192, 24, 18}, nb, true, mr == 6 ? 3 : 4, false, 20);
                fails += run_shape({384, 12, 9}, 252, true, mr == 6 ? 2 : 2, false, 20);
            }
            g_wide = 0;
            fails += run_shape({96, 48, 36}, nb, true, 4, false, 20, false);
            fails += run_shape({192, 24, 18}, nb, true, 2, false, 20, false);
            fails += run_shape({384, 12, 9}, 252, true, 1, false, 20, false);
        }
        printf(fails ? "FAILED (%d)\n" : "all shapes OK\n", fails);
        return fails ? 1 : 0;
    }
    if (mode == 0) {
        fails += run_shape({96, 48, 36}, 3, true, 2, false, 0);
        fails += run_shape({96, 48, 36}, 3, false, 1, false, 0);
        fails += run_shape({192, 24, 18}, 5, true, 3, false, 0);
        fails += run_shape({384, 12, 9}, 7, true, 1, false, 0);
        fails += run_shape({96, 16, 12}, 2, true, 1, false, 0);
    }
    g_wide = 1;
    fails += run_shape({96, 48, 36}, 3, true, 2, false, 0);
    fails += run_shape({96, 48, 36}, 3, false, 1, false, 0);
    fails += run_shape({192, 24, 18}, 5, true, 3, false, 0);
    fails += run_shape({384, 12, 9}, 7, true, 1, false, 0);
    fails += run_shape({96, 16, 12}, 2, true, 1, false, 0);
    g_wide = 0;
    for (int form = 2; form >= (mode == 0 ? 0 : 2); --form) {   // 2 wide, 1 half, 0 full
        const int half = form > 0;
        g_wide = form == 2;
        const int t = form == 1 ? 2 : 1;   // same pixels per block in all forms
        fails += run_shape({96, 48, 36}, nb, true, 2 * t, false, 20, half);
        fails += run_shape({192, 24, 18}, nb, true, 1 * t, false, 20, half);
        fails += run_shape({384, 12, 9}, nb, true, 1 * t, false, 20, half);
        fails += run_shape({96, 48, 36}, nb, true, 4 * t, false, 20, half);
        fails += run_shape({192, 24, 18}, nb, true, 2 * t, false, 20, half);
        fails += run_shape({384, 12, 9}, 252, true, 1 * t, false, 20, half);
    }
    g_wide = 0;
    if (mode == 0) {
        // co-resident half blocks out of step: 2 nb crops, 8 tiles per half block (two blocks per CU, ~180 us), start delays of
        // 0..3 x skew ticks of 10 ns (a tile of the 96-channel branch takes ~22 us)
        for (int skew : {0, 275, 550, 1100}) {
            g_skew = skew;
            printf("skew %d ticks: ", skew);
            fails += run_shape({96, 48, 36}, 2 * nb, true, 8, false, 10, true);
        }
        g_skew = 0;
        fails += run_shape({96, 48, 36}, 2 * nb, true, 4, false, 10, false);
        // one wave per SIMD: one half block per CU
        g_lds = 96 * 1024;
        printf("one block per CU: ");
        fails += run_shape({96, 48, 36}, nb, true, 8, false, 20, true);
        printf("one block per CU: ");
        fails += run_shape({192, 24, 18}, nb, true, 4, false, 20, true);
        printf("one block per CU: ");
        fails += run_shape({384, 12, 9}, 252, true, 2, false, 20, true);
        g_lds = N96H_LDS;
    }
    printf(fails ? "FAILED (%d)\n" : "all shapes OK\n", fails);
    return fails ? 1 : 0;
}
