/*
 * ORACLE (test infrastructure, NOT product code) -- plain-C restatement of the
 * reference hot path, independent of PyTorch.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product path (simple-hrnet_amd) never links or calls it.
 *
 * Restates (citations into /root/reference):
 *   HRNet.forward          models_/hrnet.py:157-189
 *   StageModule.forward    models_/hrnet.py:55-71   (fuse order j = 0..B-1)
 *   Bottleneck.forward     models_/modules.py:20-40
 *   BasicBlock.forward     models_/modules.py:56-72
 *   decode loop            SimpleHRNet.py:297-308
 *
 * The reference's arithmetic lives in PyTorch (third party, torch>=1.4,
 * requirements.txt:8; here torch 2.10 CPU / oneDNN).  Its published semantics,
 * restated here in NCHW fp32:
 *   Conv2d      zero padding k/2, no bias except final_layer, cross-correlation
 *   BatchNorm2d eval: y = (x - mean) / sqrt(var + 1e-5) * gamma + beta
 *   Upsample    nearest, integer scale s: dst[y][x] = src[y/s][x/s]
 *   np.argmax   first maximum in row-major order
 * Accumulation inside a convolution is done in double (acc_double=1) or float
 * (acc_double=0); oneDNN's own summation order is unspecified, so agreement with
 * the reference is to rounding (<= 2e-5 abs on these heat-maps), pinned by
 * tests/test_oracle.py against tests/golden/ (outputs of the reference itself).
 *
 * Build: see oracle/Makefile  (gcc -O2 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    const char *name;  /* state_dict key */
    const float *data; /* contiguous fp32, OIHW for conv weights */
} oracle_tensor;

typedef struct {
    const oracle_tensor *t;
    int n;
    int acc_double;
    int err;
} ctx_t;

static const float *find(ctx_t *cx, const char *name) {
    for (int i = 0; i < cx->n; ++i)
        if (strcmp(cx->t[i].name, name) == 0) return cx->t[i].data;
    fprintf(stderr, "hrnet_oracle: missing tensor %s\n", name);
    cx->err = 1;
    return NULL;
}

typedef struct {
    float *d;
    int n, c, h, w;
} act_t;

static act_t act_new(int n, int c, int h, int w) {
    act_t a = {(float *)malloc(sizeof(float) * (size_t)n * c * h * w), n, c, h, w};
    return a;
}
static void act_free(act_t *a) {
    free(a->d);
    a->d = NULL;
}
static size_t act_count(const act_t *a) { return (size_t)a->n * a->c * a->h * a->w; }

/* out[n][co][ho][wo] = bias[co] + sum_{ci,kh,kw} w[co][ci][kh][kw] * in[n][ci][ho*s+kh-p][wo*s+kw-p] */
static act_t conv2d(ctx_t *cx, const act_t *in, const float *w, const float *bias, int cout, int k, int stride) {
    const int pad = k / 2;
    const int ho = (in->h + 2 * pad - k) / stride + 1, wo = (in->w + 2 * pad - k) / stride + 1;
    act_t out = act_new(in->n, cout, ho, wo);
    const int cin = in->c, H = in->h, W = in->w;
    const int accd = cx->acc_double;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < in->n; ++n)
        for (int co = 0; co < cout; ++co) {
            float *o = out.d + ((size_t)n * cout + co) * ho * wo;
            for (int y = 0; y < ho; ++y)
                for (int x = 0; x < wo; ++x) {
                    double accD = 0.0;
                    float accF = 0.0f;
                    for (int ci = 0; ci < cin; ++ci) {
                        const float *ip = in->d + ((size_t)n * cin + ci) * H * W;
                        const float *wp = w + ((size_t)co * cin + ci) * k * k;
                        for (int kh = 0; kh < k; ++kh) {
                            const int iy = y * stride + kh - pad;
                            if (iy < 0 || iy >= H) continue;
                            for (int kw = 0; kw < k; ++kw) {
                                const int ix = x * stride + kw - pad;
                                if (ix < 0 || ix >= W) continue;
                                if (accd)
                                    accD += (double)wp[kh * k + kw] * (double)ip[(size_t)iy * W + ix];
                                else
                                    accF += wp[kh * k + kw] * ip[(size_t)iy * W + ix];
                            }
                        }
                    }
                    float r = accd ? (float)accD : accF;
                    if (bias) r += bias[co];
                    o[(size_t)y * wo + x] = r;
                }
        }
    return out;
}

static void bn_eval(ctx_t *cx, act_t *a, const char *prefix) {
    char key[256];
    snprintf(key, sizeof key, "%s.weight", prefix);
    const float *g = find(cx, key);
    snprintf(key, sizeof key, "%s.bias", prefix);
    const float *b = find(cx, key);
    snprintf(key, sizeof key, "%s.running_mean", prefix);
    const float *m = find(cx, key);
    snprintf(key, sizeof key, "%s.running_var", prefix);
    const float *v = find(cx, key);
    if (cx->err) return;
    const size_t hw = (size_t)a->h * a->w;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < a->n; ++n)
        for (int c = 0; c < a->c; ++c) {
            const float inv = 1.0f / sqrtf(v[c] + 1e-5f);
            float *p = a->d + ((size_t)n * a->c + c) * hw;
            for (size_t i = 0; i < hw; ++i) p[i] = (p[i] - m[c]) * inv * g[c] + b[c];
        }
}

static void relu(act_t *a) {
    const size_t cnt = act_count(a);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < cnt; ++i) a->d[i] = a->d[i] > 0.0f ? a->d[i] : 0.0f;
}

static void add_into(act_t *dst, const act_t *src) {
    const size_t cnt = act_count(dst);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < cnt; ++i) dst->d[i] = dst->d[i] + src->d[i];
}

static act_t copy_of(const act_t *a) {
    act_t o = act_new(a->n, a->c, a->h, a->w);
    memcpy(o.d, a->d, sizeof(float) * act_count(a));
    return o;
}

static act_t upsample_nearest(const act_t *a, int s) {
    act_t o = act_new(a->n, a->c, a->h * s, a->w * s);
    const int H = o.h, W = o.w;
#pragma omp parallel for schedule(static)
    for (int nc = 0; nc < a->n * a->c; ++nc)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                o.d[((size_t)nc * H + y) * W + x] = a->d[((size_t)nc * a->h + y / s) * a->w + x / s];
    return o;
}

/* conv (no bias) + BN [+ ReLU] keyed by "<conv>.weight" and "<bn>.*" */
static act_t conv_bn(ctx_t *cx, const act_t *in, const char *conv, const char *bn, int cout, int k, int stride,
                     int do_relu) {
    char key[256];
    snprintf(key, sizeof key, "%s.weight", conv);
    const float *w = find(cx, key);
    if (!w) return act_new(1, 1, 1, 1);
    act_t o = conv2d(cx, in, w, NULL, cout, k, stride);
    bn_eval(cx, &o, bn);
    if (do_relu) relu(&o);
    return o;
}

/* modules.py:20-40 */
static act_t bottleneck(ctx_t *cx, const act_t *x, const char *p, int has_down) {
    char a[256], b[256];
    snprintf(a, sizeof a, "%s.conv1", p);
    snprintf(b, sizeof b, "%s.bn1", p);
    act_t o1 = conv_bn(cx, x, a, b, 64, 1, 1, 1);
    snprintf(a, sizeof a, "%s.conv2", p);
    snprintf(b, sizeof b, "%s.bn2", p);
    act_t o2 = conv_bn(cx, &o1, a, b, 64, 3, 1, 1);
    snprintf(a, sizeof a, "%s.conv3", p);
    snprintf(b, sizeof b, "%s.bn3", p);
    act_t o3 = conv_bn(cx, &o2, a, b, 256, 1, 1, 0);
    act_free(&o1);
    act_free(&o2);
    if (has_down) {
        snprintf(a, sizeof a, "%s.downsample.0", p);
        snprintf(b, sizeof b, "%s.downsample.1", p);
        act_t r = conv_bn(cx, x, a, b, 256, 1, 1, 0);
        add_into(&o3, &r);
        act_free(&r);
    } else {
        add_into(&o3, x);
    }
    relu(&o3);
    return o3;
}

/* modules.py:56-72 */
static act_t basic_block(ctx_t *cx, const act_t *x, const char *p) {
    char a[256], b[256];
    snprintf(a, sizeof a, "%s.conv1", p);
    snprintf(b, sizeof b, "%s.bn1", p);
    act_t o1 = conv_bn(cx, x, a, b, x->c, 3, 1, 1);
    snprintf(a, sizeof a, "%s.conv2", p);
    snprintf(b, sizeof b, "%s.bn2", p);
    act_t o2 = conv_bn(cx, &o1, a, b, x->c, 3, 1, 0);
    act_free(&o1);
    add_into(&o2, x);
    relu(&o2);
    return o2;
}

/* hrnet.py:55-71; xs[0..nb) are consumed (freed), out[0..nout) are produced */
static void stage_module(ctx_t *cx, const char *p, act_t *xs, int nb, int nout, int c, act_t *out) {
    char q[256], a[300], b[300];
    for (int br = 0; br < nb; ++br)
        for (int k = 0; k < 4; ++k) {
            snprintf(q, sizeof q, "%s.branches.%d.%d", p, br, k);
            act_t y = basic_block(cx, &xs[br], q);
            act_free(&xs[br]);
            xs[br] = y;
        }
    for (int i = 0; i < nout; ++i) {
        act_t acc = {0};
        for (int j = 0; j < nb; ++j) {
            act_t t;
            snprintf(q, sizeof q, "%s.fuse_layers.%d.%d", p, i, j);
            if (i == j) {
                t = copy_of(&xs[j]);
            } else if (i < j) { /* hrnet.py:30-35 */
                snprintf(a, sizeof a, "%s.0", q);
                snprintf(b, sizeof b, "%s.1", q);
                act_t lo = conv_bn(cx, &xs[j], a, b, c << i, 1, 1, 0);
                t = upsample_nearest(&lo, 1 << (j - i));
                act_free(&lo);
            } else { /* hrnet.py:36-51 */
                t = copy_of(&xs[j]);
                for (int k = 0; k < i - j; ++k) {
                    const int last = (k == i - j - 1);
                    snprintf(a, sizeof a, "%s.%d.0", q, k);
                    snprintf(b, sizeof b, "%s.%d.1", q, k);
                    act_t n2 = conv_bn(cx, &t, a, b, last ? (c << i) : (c << j), 3, 2, !last);
                    act_free(&t);
                    t = n2;
                }
            }
            if (j == 0) {
                acc = t;
            } else { /* x_fused[i] = x_fused[i] + f(x[j]), hrnet.py:66 */
                add_into(&acc, &t);
                act_free(&t);
            }
        }
        relu(&acc);
        out[i] = acc;
    }
    for (int br = 0; br < nb; ++br) act_free(&xs[br]);
}

/* hrnet.py:157-189.  images NCHW fp32 (n,3,H,W) -> heatmaps (n,joints,H/4,W/4). returns 0 on success */
int hrnet_oracle_forward(const oracle_tensor *tensors, int ntensors, int c, int joints, const float *images, int n,
                         int H, int W, float *heatmaps, int acc_double) {
    ctx_t cx = {tensors, ntensors, acc_double, 0};
    act_t x0 = {(float *)images, n, 3, H, W};
    act_t x = conv_bn(&cx, &x0, "conv1", "bn1", 64, 3, 2, 1);
    act_t y = conv_bn(&cx, &x, "conv2", "bn2", 64, 3, 2, 1);
    act_free(&x);
    x = y;
    char p[64];
    for (int b = 0; b < 4; ++b) {
        snprintf(p, sizeof p, "layer1.%d", b);
        y = bottleneck(&cx, &x, p, b == 0);
        act_free(&x);
        x = y;
    }
    act_t xs[4], ys[4];
    xs[0] = conv_bn(&cx, &x, "transition1.0.0", "transition1.0.1", c, 3, 1, 1);
    xs[1] = conv_bn(&cx, &x, "transition1.1.0.0", "transition1.1.0.1", 2 * c, 3, 2, 1);
    act_free(&x);
    stage_module(&cx, "stage2.0", xs, 2, 2, c, ys);
    xs[0] = ys[0];
    xs[1] = ys[1];
    xs[2] = conv_bn(&cx, &xs[1], "transition2.2.0.0", "transition2.2.0.1", 4 * c, 3, 2, 1);
    for (int m = 0; m < 4; ++m) {
        snprintf(p, sizeof p, "stage3.%d", m);
        stage_module(&cx, p, xs, 3, 3, c, ys);
        for (int i = 0; i < 3; ++i) xs[i] = ys[i];
    }
    xs[3] = conv_bn(&cx, &xs[2], "transition3.3.0.0", "transition3.3.0.1", 8 * c, 3, 2, 1);
    stage_module(&cx, "stage4.0", xs, 4, 4, c, ys);
    for (int i = 0; i < 4; ++i) xs[i] = ys[i];
    stage_module(&cx, "stage4.1", xs, 4, 4, c, ys);
    for (int i = 0; i < 4; ++i) xs[i] = ys[i];
    stage_module(&cx, "stage4.2", xs, 4, 1, c, ys);
    const float *fw = find(&cx, "final_layer.weight");
    const float *fb = find(&cx, "final_layer.bias");
    if (!cx.err) {
        act_t hm = conv2d(&cx, &ys[0], fw, fb, joints, 1, 1);
        memcpy(heatmaps, hm.d, sizeof(float) * act_count(&hm));
        act_free(&hm);
    }
    act_free(&ys[0]);
    return cx.err;
}

/* SimpleHRNet.py:297-308.  boxes are [x1,y1,x2,y2]; box_is_int selects int32 (multi-person, line 230)
 * or float32 (single-person, line 223) storage.  pts: (n,joints,3) = (y, x, confidence). */
void hrnet_oracle_decode(const float *heatmaps, int n, int joints, int h, int w, const void *boxes, int box_is_int,
                         float *pts) {
    for (int i = 0; i < n; ++i) {
        double bx1, by1, dx, dy;
        if (box_is_int) {
            const int32_t *b = (const int32_t *)boxes + 4 * (size_t)i;
            bx1 = b[0], by1 = b[1], dx = (double)(b[2] - b[0]), dy = (double)(b[3] - b[1]);
        } else {
            const float *b = (const float *)boxes + 4 * (size_t)i;
            bx1 = b[0], by1 = b[1], dx = (double)(b[2] - b[0]), dy = (double)(b[3] - b[1]); /* fp32 subtract */
        }
        for (int j = 0; j < joints; ++j) {
            const float *hm = heatmaps + ((size_t)i * joints + j) * h * w;
            int best = 0;
            for (int k = 1; k < h * w; ++k)
                if (hm[k] > hm[best]) best = k; /* strict >: first maximum wins, as np.argmax */
            const int py = best / w, px = best % w;
            float *o = pts + ((size_t)i * joints + j) * 3;
            o[0] = (float)((double)py * 1. / (double)h * dy + by1);
            o[1] = (float)((double)px * 1. / (double)w * dx + bx1);
            o[2] = hm[best];
        }
    }
}
