// Stand-alone check + timing of the persistent work-queue form of the grouped BasicBlock launch
// (simple-hrnet_amd/csrc/conv3x3_queue.inc) against a naive GPU convolution on the same flat padded NHWC tensors, and against
// the per-block form (conv3x3_n96.inc) on the same unit list (bit-identical, and the time of both).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/c3q_test tools/c3q_test.hip
//   tools/bin/c3q_test [crops]
#include "../simple-hrnet_amd/csrc/conv3x3_lds.hip"
namespace hrn {
int conv3x3_f32_bm(int) { return 0; }
hipError_t launch_conv3x3_f32(const Conv3Problem *, const void *, int, int, int, hipStream_t) { return hipErrorInvalidValue; }
}  // namespace hrn
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <map>
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

struct Conv {   // one convolution with its tensors
    int C, H, W, wp, hpwp, m, relu;
    bool with_res;
    size_t rows;
    int guard_front;
    uint16_t *din, *dres, *dout, *dout2, *dref, *dw, *dwref;
    float *dbias;
    Conv3Problem p;
};

static Conv make_conv(int C, int H, int W, int nb, bool with_res, int relu, int seed) {
    Conv c{};
    c.C = C, c.H = H, c.W = W, c.wp = W + 1, c.hpwp = (H + 1) * (W + 1), c.m = nb * c.hpwp, c.relu = relu, c.with_res = with_res;
    c.guard_front = c.wp + 1;
    c.rows = (size_t)c.guard_front + c.m + c.wp + 1 + 512;
    std::vector<uint16_t> hin(c.rows * C, 0), hres(c.rows * C, 0);
    srand(seed);
    for (int n = 0; n < nb; ++n)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                for (int ch = 0; ch < C; ++ch) {
                    const size_t r = (size_t)c.guard_front + (size_t)n * c.hpwp + y * c.wp + x;
                    hin[r * C + ch] = f2bf_h((rand() % 2001 - 1000) / 1000.f);
                    hres[r * C + ch] = f2bf_h((rand() % 2001 - 1000) / 500.f);
                }
    const int K = 9 * C;
    std::vector<float> wf((size_t)C * K);
    std::vector<uint16_t> wref((size_t)C * K);
    for (size_t i = 0; i < wf.size(); ++i) {
        wf[i] = (rand() % 2001 - 1000) / 1000.f / sqrtf((float)K) * 2.f;
        wref[i] = f2bf_h(wf[i]);
    }
    std::vector<float> hb(C);
    for (int ch = 0; ch < C; ++ch) hb[ch] = (rand() % 2001 - 1000) / 2000.f;
    const int KS = 32, NRB = 6, slices = C / KS, ntiles = C / 96, nch = 9;
    std::vector<uint16_t> wpk((size_t)ntiles * slices * nch * NRB * 512);
    for (int t = 0; t < ntiles; ++t)
        for (int s = 0; s < slices; ++s) {
            uint16_t *blk = wpk.data() + ((size_t)t * slices + s) * nch * NRB * 512;
            for (int cc = 0; cc < nch; ++cc)
                for (int j = 0; j < NRB; ++j)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int li = lane & 15, g = lane >> 4;
                        const int co = t * 96 + (j >> 1) * 32 + (li >> 2) * 8 + (j & 1) * 4 + (li & 3);
                        uint16_t *d = blk + ((size_t)(cc * NRB + j) * 64 + lane) * 8;
                        for (int e = 0; e < 8; ++e) {
                            const int kl = 32 * cc + 8 * g + e;
                            const int tap = kl / KS, cil = kl % KS;
                            d[e] = f2bf_h(wf[(size_t)co * K + tap * C + s * KS + cil]);
                        }
                    }
        }
    hipMalloc(&c.din, c.rows * C * 2), hipMalloc(&c.dres, c.rows * C * 2), hipMalloc(&c.dout, c.rows * C * 2), hipMalloc(&c.dout2, c.rows * C * 2),
        hipMalloc(&c.dref, c.rows * C * 2);
    hipMalloc(&c.dw, wpk.size() * 2), hipMalloc(&c.dwref, wref.size() * 2), hipMalloc(&c.dbias, C * 4);
    hipMemcpy(c.din, hin.data(), c.rows * C * 2, hipMemcpyHostToDevice);
    hipMemcpy(c.dres, hres.data(), c.rows * C * 2, hipMemcpyHostToDevice);
    hipMemset(c.dout, 0x7f, c.rows * C * 2), hipMemset(c.dout2, 0x7f, c.rows * C * 2), hipMemset(c.dref, 0, c.rows * C * 2);
    hipMemcpy(c.dw, wpk.data(), wpk.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(c.dwref, wref.data(), wref.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(c.dbias, hb.data(), C * 4, hipMemcpyHostToDevice);
    Conv3Problem &p = c.p;
    memset(&p, 0, sizeof p);
    const size_t gf = (size_t)c.guard_front * C;
    p.in = c.din + gf, p.out = c.dout + gf, p.w = c.dw, p.bias = c.dbias;
    p.res = with_res ? c.dres + gf : nullptr;
    p.cin = C, p.cout = C, p.h = H, p.wd = W, p.wp = c.wp, p.hpwp = c.hpwp, p.relu = relu, p.slices = slices, p.ntiles = ntiles;
    p.tiles_per_block = 1, p.bm = conv3x3_lds_bm(32, 6, c.wp);
    fast_div_h(c.hpwp, &p.magic_hpwp, &p.shift_hpwp), fast_div_h(c.wp, &p.magic_wp, &p.shift_wp);
    p.n96 = 1;
    return c;
}
static void free_conv(Conv &c) {
    hipFree(c.din), hipFree(c.dres), hipFree(c.dout), hipFree(c.dout2), hipFree(c.dref), hipFree(c.dw), hipFree(c.dwref), hipFree(c.dbias);
}

struct Ent {
    double key;
    int prob, nt, tiles, mt0;
};

// convs: (C, H, W, tiles per unit); one grouped launch over all of them
static int run_group(const char *name, std::vector<Conv> &cv, const std::vector<int> &tpb, int nb, int reps, int nblocks = 256) {
    // unit list: per convolution the XCD-aware order of the library (rounds of 8 M groups x all cout tiles), longest first
    std::vector<Ent> ents;
    for (size_t k = 0; k < cv.size(); ++k) {
        if (cv[k].p.bm != 512) {
            printf("%s: conv %zu does not take 512-pixel tiles\n", name, k);
            return 1;
        }
        const int mtiles = (cv[k].m + 511) / 512, mgroups = (mtiles + tpb[k] - 1) / tpb[k];
        int seq = 0;
        for (int round = 0; round * 8 < mgroups; ++round)
            for (int nt = 0; nt < cv[k].p.ntiles; ++nt)
                for (int x = 0; x < 8; ++x) {
                    const int mg = round * 8 + x;
                    if (mg >= mgroups) continue;
                    const int tiles = std::min(tpb[k], mtiles - mg * tpb[k]);
                    ents.push_back({-(double)tiles * cv[k].p.slices + 1e-6 * seq++, (int)k, nt, tiles, mg * tpb[k]});
                }
    }
    std::stable_sort(ents.begin(), ents.end(), [](const Ent &a, const Ent &b) { return a.key < b.key; });
    std::vector<QUnit> units;
    std::vector<int2> map;
    std::vector<Conv3Problem> probs, probs2;
    for (auto &c : cv) {
        probs.push_back(c.p);
        Conv3Problem q = c.p;
        q.out = c.dout2 + (size_t)c.guard_front * c.C;
        probs2.push_back(q);
    }
    for (const Ent &e : ents) {
        units.push_back(make_qunit(cv[e.prob].p, e.nt, e.mt0, e.tiles, nb));
        map.push_back(int2{e.prob | (e.nt << 8) | (e.tiles << 16), e.mt0});
    }
    QUnit *dunits;
    int2 *dmap;
    int *dheads;
    Conv3Problem *dprobs, *dprobs2;
    hipMalloc(&dunits, units.size() * sizeof(QUnit)), hipMalloc(&dmap, map.size() * sizeof(int2)), hipMalloc(&dheads, 64);
    hipMalloc(&dprobs, probs.size() * sizeof(Conv3Problem)), hipMalloc(&dprobs2, probs.size() * sizeof(Conv3Problem));
    hipMemcpy(dunits, units.data(), units.size() * sizeof(QUnit), hipMemcpyHostToDevice);
    hipMemcpy(dmap, map.data(), map.size() * sizeof(int2), hipMemcpyHostToDevice);
    hipMemcpy(dprobs, probs.data(), probs.size() * sizeof(Conv3Problem), hipMemcpyHostToDevice);
    hipMemcpy(dprobs2, probs2.data(), probs.size() * sizeof(Conv3Problem), hipMemcpyHostToDevice);
    hipMemset(dheads, 0, 64);
    hipError_t e = launch_conv3x3_queue(dunits, (int)units.size(), dheads, dprobs, 0, 0, 0, 0, nb, nblocks, 0);
    hipError_t e2 = hipDeviceSynchronize();
    if (e != hipSuccess || e2 != hipSuccess) {
        printf("%s: queue launch failed: %s / %s\n", name, hipGetErrorString(e), hipGetErrorString(e2));
        return 1;
    }
    int heads[8];
    hipMemcpy(heads, dheads, 32, hipMemcpyDeviceToHost);
    // the per-block form on the same unit list, into the second output tensors
    e = launch_conv3x3_lds(dprobs2, dmap, (int)map.size(), nb, 32, 6, 0);
    e2 = hipDeviceSynchronize();
    if (e != hipSuccess || e2 != hipSuccess) {
        printf("%s: per-block launch failed: %s / %s\n", name, hipGetErrorString(e), hipGetErrorString(e2));
        return 1;
    }
    long bad = 0, diff = 0, tail_bad = 0;
    double maxerr = 0;
    double gflop = 0;
    for (auto &c : cv) {
        const long total = (long)c.m * c.C;
        const size_t gf = (size_t)c.guard_front * c.C;
        ref_conv<<<(unsigned)((total + 255) / 256), 256>>>(c.din + gf, c.dwref, c.dbias, c.with_res ? c.dres + gf : nullptr, c.dref + gf, c.m, c.C, c.C,
                                                           c.H, c.W, c.wp, c.hpwp, c.relu);
        hipDeviceSynchronize();
        std::vector<uint16_t> ho(c.rows * c.C), ho2(c.rows * c.C), hr(c.rows * c.C);
        hipMemcpy(ho.data(), c.dout, c.rows * c.C * 2, hipMemcpyDeviceToHost);
        hipMemcpy(ho2.data(), c.dout2, c.rows * c.C * 2, hipMemcpyDeviceToHost);
        hipMemcpy(hr.data(), c.dref, c.rows * c.C * 2, hipMemcpyDeviceToHost);
        long first_bad = -1;
        for (long q = 0; q < c.m; ++q)
            for (int ch = 0; ch < c.C; ++ch) {
                const size_t i = ((size_t)c.guard_front + q) * c.C + ch;
                const float a = bf2f_h(ho[i]), b = bf2f_h(hr[i]);
                const float err = fabsf(a - b);
                if (!(err <= 0.02f + 0.01f * fabsf(b))) {
                    if (first_bad < 0) first_bad = q * c.C + ch;
                    ++bad;
                }
                if (err > maxerr) maxerr = err;
                if (ho[i] != ho2[i]) ++diff;
            }
        for (size_t i = ((size_t)c.guard_front + c.m) * c.C; i < c.rows * c.C; ++i)
            if (ho[i] != 0 && ho[i] != 0x7f7f) ++tail_bad;
        for (size_t i = 0; i < (size_t)c.guard_front * c.C; ++i)
            if (ho[i] != 0x7f7f) ++tail_bad;
        if (first_bad >= 0) {
            const long q = first_bad / c.C;
            printf("   C=%d first bad: row %ld (img %ld, y %ld, x %ld, M tile %ld) ch %ld\n", c.C, q, q / c.hpwp, (q % c.hpwp) / c.wp, (q % c.hpwp) % c.wp,
                   q / 512, first_bad % c.C);
        }
        gflop += 2.0 * 9 * c.C * (double)c.C * c.H * c.W * nb * 1e-9;
    }
    printf("%-34s nb=%3d units=%5zu blocks=%3d : bad=%ld differs-from-per-block=%ld tail_bad=%ld maxerr=%.4f heads=%d,%d,%d,%d,..", name, nb, units.size(),
           nblocks, bad, diff, tail_bad, maxerr, heads[0], heads[1], heads[2], heads[3]);
    if (reps > 0) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        float msq = 0, msb = 0;
        for (int i = 0; i < 3; ++i) {
            hipMemsetAsync(dheads, 0, 32, 0);
            launch_conv3x3_queue(dunits, (int)units.size(), dheads, dprobs, 0, 0, 0, 0, nb, nblocks, 0);
            launch_conv3x3_lds(dprobs2, dmap, (int)map.size(), nb, 32, 6, 0);
        }
        // interleaved rounds: queue form (with its memset), per-block form
        float bq = 1e30f, bb = 1e30f;
        for (int r = 0; r < reps; ++r) {
            hipEventRecord(e0);
            hipMemsetAsync(dheads, 0, 32, 0);
            launch_conv3x3_queue(dunits, (int)units.size(), dheads, dprobs, 0, 0, 0, 0, nb, nblocks, 0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            msq += ms, bq = std::min(bq, ms);
            hipEventRecord(e0);
            launch_conv3x3_lds(dprobs2, dmap, (int)map.size(), nb, 32, 6, 0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            msb += ms, bb = std::min(bb, ms);
        }
        printf("\n      queue %.1f us (best %.1f) = %.0f TFLOP/s   per-block %.1f us (best %.1f) = %.0f TFLOP/s   ratio %.3f", msq / reps * 1e3, bq * 1e3,
               gflop / (msq / reps), msb / reps * 1e3, bb * 1e3, gflop / (msb / reps), msb / msq);
    }
    printf("\n");
#ifdef HRN_Q_TIMING
    if (reps > 0) {   // when do the blocks of one queue launch finish?  (s_memrealtime, 10 ns ticks)
        long long *dt;
        hipMalloc(&dt, nblocks * 32);
        hipMemset(dt, 0, nblocks * 32);
        hipMemcpyToSymbol(HIP_SYMBOL(g_q_timing), &dt, sizeof(dt));
        hipMemsetAsync(dheads, 0, 32, 0);
        launch_conv3x3_queue(dunits, (int)units.size(), dheads, dprobs, 0, 0, 0, 0, nb, nblocks, 0);
        hipDeviceSynchronize();
        std::vector<long long> ht(nblocks * 4);
        hipMemcpy(ht.data(), dt, nblocks * 32, hipMemcpyDeviceToHost);
        long long t0 = ht[0], tmax = 0;
        for (int b = 0; b < nblocks; ++b) t0 = std::min(t0, ht[b * 4]);
        std::vector<double> ends;
        double sum = 0;
        for (int b = 0; b < nblocks; ++b) ends.push_back((ht[b * 4 + 2] - t0) * 0.01), sum += ends.back(), tmax = std::max(tmax, ht[b * 4 + 2] - t0);
        std::sort(ends.begin(), ends.end());
        printf("      block finish times (us after the first start): min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f  mean %.1f  -> tail idle %.1f %% of CU time\n",
               ends.front(), ends[nblocks / 10], ends[nblocks / 2], ends[nblocks * 9 / 10], ends.back(), sum / nblocks, 100.0 * (1.0 - sum / nblocks / ends.back()));
        // the per-block form on the same list: per CU (XCC + HW_ID bits), when its blocks ran
        {
            const int nbk = (int)map.size();
            long long *db;
            hipMalloc(&db, (size_t)nbk * 32);
            hipMemset(db, 0, (size_t)nbk * 32);
            hipMemcpyToSymbol(HIP_SYMBOL(g_q_timing), &db, sizeof(db));
            launch_conv3x3_lds(dprobs2, dmap, nbk, nb, 32, 6, 0);
            hipDeviceSynchronize();
            std::vector<long long> hb((size_t)nbk * 4);
            hipMemcpy(hb.data(), db, (size_t)nbk * 32, hipMemcpyDeviceToHost);
            std::map<long long, std::vector<std::pair<long long, long long>>> per_cu;
            long long b0 = hb[0], bend = 0;
            for (int b = 0; b < nbk; ++b) b0 = std::min(b0, hb[b * 4]), bend = std::max(bend, hb[b * 4 + 2]);
            for (int b = 0; b < nbk; ++b) per_cu[hb[b * 4 + 3]].push_back({hb[b * 4] - b0, hb[b * 4 + 2] - b0});
            double busy = 0, gaps = 0, last_sum = 0, first_sum = 0;
            std::vector<double> lasts;
            long ngaps = 0;
            for (auto &kv : per_cu) {
                auto &v = kv.second;
                std::sort(v.begin(), v.end());
                first_sum += v.front().first * 0.01;
                for (size_t i = 0; i < v.size(); ++i) {
                    busy += (v[i].second - v[i].first) * 0.01;
                    if (i) gaps += (v[i].first - v[i - 1].second) * 0.01, ++ngaps;
                }
                lasts.push_back(v.back().second * 0.01), last_sum += lasts.back();
            }
            std::sort(lasts.begin(), lasts.end());
            const double total = (bend - b0) * 0.01, ncu = (double)per_cu.size();
            printf("      per-block form: %zu CUs seen, launch %.1f us; per CU: first block starts at %.2f us, busy %.1f us in %.1f blocks, gaps between blocks %.2f us "
                   "(%.2f us each), last block ends: median %.1f  p10 %.1f  max %.1f -> tail idle %.1f %%, gaps %.1f %% of CU time\n",
                   per_cu.size(), total, first_sum / ncu, busy / ncu, nbk / ncu, gaps / ncu, ngaps ? gaps / ngaps : 0.0, lasts[lasts.size() / 2],
                   lasts[lasts.size() / 10], lasts.back(), 100.0 * (1.0 - last_sum / ncu / lasts.back()), 100.0 * gaps / ncu / lasts.back());
            hipFree(db);
        }
        long long *nul = nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(g_q_timing), &nul, sizeof(nul));
        hipFree(dt);
    }
#endif
    hipFree(dunits), hipFree(dmap), hipFree(dheads), hipFree(dprobs), hipFree(dprobs2);
    return bad || diff || tail_bad ? 1 : 0;
}

int main(int argc, char **argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 256;
    int fails = 0;
    {   // small ragged cases: one conv, a handful of units (most blocks find nothing; some get one, two, three units)
        std::vector<Conv> cv = {make_conv(96, 48, 36, 3, true, 1, 11)};
        fails += run_group("96ch 3 crops tpb 2", cv, {2}, 3, 0);
        for (auto &c : cv) hipMemset(c.dout, 0x7f, c.rows * c.C * 2);
        fails += run_group("96ch 3 crops tpb 1, 2 blocks", cv, {1}, 3, 0, 2);    // two blocks work the whole queue (stealing from every list)
        for (auto &c : cv) hipMemset(c.dout, 0x7f, c.rows * c.C * 2);
        fails += run_group("96ch 3 crops tpb 1, 1 block", cv, {1}, 3, 0, 1);
        for (auto &c : cv) free_conv(c);
    }
    {   // three convolutions of different widths / grids in one launch, few blocks: units of different geometry follow each other
        std::vector<Conv> cv = {make_conv(96, 48, 36, 5, true, 1, 21), make_conv(192, 24, 18, 5, false, 1, 22), make_conv(384, 12, 9, 5, true, 0, 23)};
        fails += run_group("96+192+384 5 crops, 8 blocks", cv, {2, 1, 1}, 5, 0, 8);
        for (auto &c : cv) hipMemset(c.dout, 0x7f, c.rows * c.C * 2);
        fails += run_group("96+192+384 5 crops, 3 blocks", cv, {1, 1, 1}, 5, 0, 3);
        for (auto &c : cv) hipMemset(c.dout, 0x7f, c.rows * c.C * 2);
        fails += run_group("96+192+384 5 crops, 256 blocks", cv, {1, 2, 1}, 5, 0, 256);
        for (auto &c : cv) free_conv(c);
    }
    {   // full size: the three wide branches of a stage-4 module at nb crops, one launch; unit lengths as the library's short / long blocks
        std::vector<Conv> cv = {make_conv(96, 48, 36, nb, true, 1, 31), make_conv(192, 24, 18, nb, true, 1, 32), make_conv(384, 12, 9, nb, true, 1, 33)};
        fails += run_group("stage-4 wide branches, tpb 2/1/1", cv, {2, 1, 1}, nb, 10);
        for (auto &c : cv) hipMemset(c.dout, 0x7f, c.rows * c.C * 2);
        fails += run_group("stage-4 wide branches, tpb 1/1/1", cv, {1, 1, 1}, nb, 10);
        for (auto &c : cv) hipMemset(c.dout, 0x7f, c.rows * c.C * 2);
        fails += run_group("stage-4 wide branches, tpb 4/2/1", cv, {4, 2, 1}, nb, 10);
        for (auto &c : cv) free_conv(c);
    }
    printf(fails ? "FAILED (%d)\n" : "all cases OK\n", fails);
    return fails ? 1 : 0;
}
