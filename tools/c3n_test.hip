// Stand-alone check + timing of the 96-cout form of the BasicBlock convolution (simple-hrnet_amd/csrc/conv3x3_n96.inc)
// against a naive GPU convolution on the same flat padded NHWC tensors.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I simple-hrnet_amd/csrc -o /tmp/c3n_test tools/c3n_test.hip
//   /tmp/c3n_test [crops]            (checks every output element of each shape, then times 20 launches)
#include "../simple-hrnet_amd/csrc/conv3x3_lds.hip"
// the library routes ks = 16 to the fp32 kernel (conv3x3_f32.hip), which this stand-alone harness does not link
namespace hrn {
int conv3x3_f32_bm(int) { return 0; }
hipError_t launch_conv3x3_f32(const Conv3Problem *, const void *, int, int, int, hipStream_t) { return hipErrorInvalidValue; }
}  // namespace hrn
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

struct Shape {
    int c, h, w;
};

static int run_shape(const Shape &sh, int nb, bool with_res, int tpb, bool small, int reps) {
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
    const int bm = small ? 128 : p.bm;
    const int mtiles = (m + bm - 1) / bm;
    std::vector<int2> map;
    const int mgroups = (mtiles + tpb - 1) / tpb;
    for (int round = 0; round * 8 < mgroups; ++round)
        for (int nt = 0; nt < ntiles; ++nt)
            for (int x = 0; x < 8; ++x) {
                const int mg = round * 8 + x;
                if (mg >= mgroups) continue;
                const int tiles = std::min(tpb, mtiles - mg * tpb);
                map.push_back(int2{0 | (nt << 8) | (tiles << 16), (mg * tpb) | (small ? 1 << 30 : 0)});
            }
    Conv3Problem *dp;
    int2 *dmap;
    hipMalloc(&dp, sizeof p), hipMalloc(&dmap, map.size() * sizeof(int2));
    hipMemcpy(dp, &p, sizeof p, hipMemcpyHostToDevice);
    hipMemcpy(dmap, map.data(), map.size() * sizeof(int2), hipMemcpyHostToDevice);
    hipError_t e = launch_conv3x3_lds(dp, dmap, (int)map.size(), nb, 32, 6, 0);
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
    // timing
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float ms = 0;
    if (reps > 0) {
        for (int i = 0; i < 3; ++i) launch_conv3x3_lds(dp, dmap, (int)map.size(), nb, 32, 6, 0);
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) launch_conv3x3_lds(dp, dmap, (int)map.size(), nb, 32, 6, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        ms /= reps;
    }
    const double gflop = 2.0 * 9 * C * (double)C * H * W * nb * 1e-9;
    printf("C=%3d %3dx%-3d nb=%3d res=%d tpb=%d bm=%d blocks=%5zu : bad=%ld (first %ld) tail_bad=%ld maxerr=%.4f", C, H, W, nb, (int)with_res, tpb,
           bm, map.size(), bad, first_bad, tail_bad, maxerr);
    if (reps > 0) {
        const int rounds = ((int)map.size() + 255) / 256;
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
    int fails = 0;
    // small, ragged cases first (every tile position, partial last tile, 128-pixel tiles)
    fails += run_shape({96, 48, 36}, 3, true, 2, false, 0);
    fails += run_shape({96, 48, 36}, 3, false, 1, true, 0);
    fails += run_shape({192, 24, 18}, 5, true, 3, false, 0);
    fails += run_shape({384, 12, 9}, 7, true, 1, false, 0);
    fails += run_shape({96, 16, 12}, 2, true, 1, true, 0);
    // full size (BASELINE configs[2]: the three wide branches at 256 crops)
    fails += run_shape({96, 48, 36}, nb, true, 2, false, 20);
    fails += run_shape({192, 24, 18}, nb, true, 1, false, 20);
    fails += run_shape({384, 12, 9}, nb, true, 1, false, 20);
    fails += run_shape({96, 48, 36}, nb, false, 4, false, 20);
    // one block per CU, long blocks: the steady-state rate of a stage
    fails += run_shape({96, 48, 36}, nb, true, 4, false, 20);
    fails += run_shape({192, 24, 18}, nb, true, 2, false, 20);
    fails += run_shape({384, 12, 9}, 252, true, 1, false, 20);
    printf(fails ? "FAILED (%d)\n" : "all shapes OK\n", fails);
    return fails ? 1 : 0;
}

#ifdef N96_PROBE   // register need of each instantiation on its own (-Rpass-analysis=kernel-resource-usage)
template <int MR>
__global__ __launch_bounds__(512, 2) void n96_probe(const hrn::Conv3Problem *probs, int nt, int mt0, int tiles, int nb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    hrn::c3n_run<MR>(probs[0], nt, mt0, tiles, nb, smem);
}
template __global__ void n96_probe<4>(const hrn::Conv3Problem *, int, int, int, int);
template __global__ void n96_probe<3>(const hrn::Conv3Problem *, int, int, int, int);
template __global__ void n96_probe<1>(const hrn::Conv3Problem *, int, int, int, int);
#endif
