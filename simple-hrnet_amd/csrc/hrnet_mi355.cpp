// Host runtime of the MI355X-native HRNet hot path: graph compiler (static launch list for a given
// width c / resolution), BatchNorm folding + MFMA-fragment weight packing, workspace planner and the
// C ABI declared in include/hrnet_mi355.h.  Graph follows models_/hrnet.py:157-189 (HRNet.forward),
// :55-71 (StageModule.forward) and models_/modules.py:20-40,56-72 of the reference.
#include "../../include/hrnet_mi355.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "kernels.h"

namespace {
#include "ctx_types.h"
}  // namespace

namespace {
// hrn_env() (kernels.h) + a record of what was found: hrn_create copies it into the handle (hrn_switches)
thread_local std::string g_env_seen;
const char *env_sw(const char *name) {
    const char *v = hrn_env(name);
    if (v) {
        const std::string item = std::string(name) + "=" + v + ";";
        if (g_env_seen.find(item) == std::string::npos) g_env_seen += item;
    }
    return v;
}
}  // namespace

struct hrn_ctx {
    std::string switches;   // the HRN_* variables this handle saw when it was created ("NAME=value;...")
    int c, joints, H, W, dtype, max_batch, device;
    int model = 0;  // 0 = HRNet (c = width), 1 = PoseResNet (c = ResNet size), SimpleHRNet.py:109-112
    int head_c = 0; // channels of the tensor final_layer reads
    int pool_in_t = -1, pool_out_t = -1;
    bool plan_only;
    int esize;
    std::string err;

    std::vector<Tensor> tensors;
    std::vector<Buffer> buffers;
    std::vector<ConvOp> convs;
    std::vector<FuseOp> fuses;
    std::vector<Conv3Group> groups;
    std::vector<DirectGroup> dgroups;
    std::vector<S2Group> s2groups;
    S2Problem *s2probs_dev = nullptr;   // (the problems of `s2groups`, then of `stemf`)
    struct Chain {
        int conv3, conv1, ds;  // conv3 of Bottleneck b, conv1 of Bottleneck b+1 (-1: none, the layer's last block), projection shortcut folded in (or -1)
        int conv2 = -1;        // round 5: the 3x3 conv of Bottleneck b computed in front of conv3 (or -1: it has its own launch)
    };
    std::vector<Chain> chains;
    std::vector<TapPoint> taps;
    Conv3Problem *probs_dev = nullptr;
    std::vector<Op> ops;
    int stem_out_t = -1, head_in_t = -1;
    int64_t stem_w_off = 0, stem_b_off = 0, stem_wp_off = 0, head_w_off = 0, head_b_off = 0, head_wp_off = 0;

    int64_t blob_bytes = 0;
    char *blob = nullptr;  // device (or host when plan_only)
    bool weights_loaded = false;

    bool disable_lds = env_sw("HRN_DISABLE_LDS") != nullptr;
    bool disable_group = env_sw("HRN_DISABLE_GROUP") != nullptr;
    bool disable_dgroup = env_sw("HRN_DISABLE_DGROUP") != nullptr;
    bool disable_chain = env_sw("HRN_DISABLE_CHAIN") != nullptr;
    bool small_tiles = env_sw("HRN_SMALL_TILES") ? atoi(env_sw("HRN_SMALL_TILES")) != 0 : true;
    int small_below = env_sw("HRN_SMALL_BELOW") ? atoi(env_sw("HRN_SMALL_BELOW")) : 384;
    int small_keep_above = env_sw("HRN_SMALL_KEEP") ? atoi(env_sw("HRN_SMALL_KEEP")) : 300;   // (0: every convolution of a small launch takes 128-pixel tiles, round 5's rule)
    bool disable_chain_ds = env_sw("HRN_DISABLE_CHAIN_DS") != nullptr;
    // generic conv kernel, bf16: the block's weights through LDS instead of one copy per wave from L2 (+1.6 % on the pass)
    // residual-prefetch variant of the K = 64 1x1 convs; persistent blocks of a chain-kernel launch (read here, at create,
    // like every other switch -- not in a function-local static at the first launch)
    int pre_mode = env_sw("HRN_PRE_MODE") ? atoi(env_sw("HRN_PRE_MODE")) : 1;
    int chain_blocks = env_sw("HRN_CHAIN_BLOCKS") ? atoi(env_sw("HRN_CHAIN_BLOCKS")) : 512;
    // round 5: the 3x3 conv of a Bottleneck without projection shortcut in front of its conv3 inside the chain kernel (off: its own launch)
    bool disable_chain3 = env_sw("HRN_DISABLE_CHAIN3") != nullptr;
    bool direct_wlds = !(env_sw("HRN_DIRECT_WLDS") && atoi(env_sw("HRN_DIRECT_WLDS")) == 0);
    // round 6: stride-2 3x3 convolutions with cin % 32 == 0 on the generic kernel request their pixel fragments two K chunks ahead by
    // LDS-DMA into a per-wave LDS ring (kernels.hip: XL); same arithmetic in the same order: bit-identical to HRN_DIRECT_XLDS=0
    bool direct_xlds = !(env_sw("HRN_DIRECT_XLDS") && atoi(env_sw("HRN_DIRECT_XLDS")) == 0);
    bool conv_xl(const ConvOp &cv) const {
        return direct_xlds && direct_wlds && dtype == 1 && cv.k == 3 && cv.stride == 2 && cv.cin % 32 == 0 && !cv.up && cv.res_t < 0 && cv.nr == 6 &&
               cv.kchunks >= 4;
    }
    // fused BasicBlocks on the 48-channel branch (conv3x3_lds.hip: bbf_run): bit-identical, 2.5x less HBM traffic on that
    // branch, +2.6 % on the whole pass at 256 crops; HRN_BBF=0 goes back to two launches per block
    bool disable_bbf = env_sw("HRN_BBF") && atoi(env_sw("HRN_BBF")) == 0;
    int bbf_tpb_div = env_sw("HRN_BBF_TPB_DIV") ? std::max(1, atoi(env_sw("HRN_BBF_TPB_DIV"))) : 3;  // a fused tile ~ 3 plain ones
    // fused only when the call has at least this many 512-pixel tiles (four per CU): measured at 384x288 +2 % at 256
    // crops (3541 tiles), +1 % at 128-192, +6 % at 96 (1328 tiles), -1 % at 64 (885 tiles), -2 % at 20, -5 % at one crop,
    // where the two plain launches with their smaller tiles spread the work over more CUs
    int bbf_min_tiles = env_sw("HRN_BBF_MIN_TILES") ? atoi(env_sw("HRN_BBF_MIN_TILES")) : 1100;
    // 1: convolutions that read the same tensor share one cout-tile width so that they can share a launch (and L2)
    int dgroup_nr_mode = env_sw("HRN_DGROUP_NR") ? atoi(env_sw("HRN_DGROUP_NR")) : 1;
    bool disable_lds32 = env_sw("HRN_DISABLE_LDS32") != nullptr;
    bool disable_n96 = env_sw("HRN_DISABLE_N96") != nullptr;
    bool disable_fgroup = env_sw("HRN_DISABLE_FGROUP") != nullptr;   // one launch per fuse output instead of one per StageModule
    bool disable_f32lds = env_sw("HRN_DISABLE_F32LDS") != nullptr;   // fp32 3x3 stride-1 convs back on the generic kernel
    int f32_small_slices = env_sw("HRN_F32_SMALL_SLICES") ? atoi(env_sw("HRN_F32_SMALL_SLICES")) : 0;   // fp32: 128-pixel tiles from this many slices on (0: never; 4 and 8 measured: no gain)
    // stride-2 slab kernel (conv_s2.hip) off: those convolutions stay on the generic kernel (bit-identical results)
    std::vector<Conv3Problem> probs_host;
    // round 4: the 96-cout form enumerates real pixels only (no MFMAs on the pad column / pad row of the flat layout) on grids whose
    // padding is at least this share of the flat pixels; bit-identical to the flat enumeration
    bool disable_compact = env_sw("HRN_DISABLE_COMPACT") != nullptr;
    double compact_min_pad = env_sw("HRN_COMPACT_MIN_PAD") ? atof(env_sw("HRN_COMPACT_MIN_PAD")) : 0.08;
    bool disable_s2 = env_sw("HRN_DISABLE_S2") != nullptr;
    // the slab kernel is taken when a launch has at least this many tiles (one per CU); smaller calls use the generic kernel
    int s2_min_tiles = env_sw("HRN_S2_MIN_TILES") ? atoi(env_sw("HRN_S2_MIN_TILES")) : 256;
    int s2_target_blocks = env_sw("HRN_S2_BLOCKS") ? std::max(1, atoi(env_sw("HRN_S2_BLOCKS"))) : 256;
    bool disable_stem_mfma = env_sw("HRN_DISABLE_STEM_MFMA") != nullptr;
    // round 4: conv1 + conv2 of the stem as one kernel (stem_fused.hip), bit-identical to the two launches it replaces
    bool disable_stem_fuse = env_sw("HRN_DISABLE_STEM_FUSE") != nullptr;
    bool stem_fuse = false;       // planned (bf16 HRNet whose geometry fits: stem_fused_fits)
    int stem_conv2_op = -1;       // ops[] index of conv2's own launch (skipped by a pass that ran the fused kernel)
    S2Group stemf;                // the fused kernel's problem (conv2 with one output row per tile) and block maps
    bool disable_head_mfma = env_sw("HRN_DISABLE_HEAD_MFMA") != nullptr;
    bool direct_nr6 = env_sw("HRN_DIRECT_NR6") ? atoi(env_sw("HRN_DIRECT_NR6")) != 0 : true;
    int half_stages_per_block = env_sw("HRN_HALF_STAGES") ? atoi(env_sw("HRN_HALF_STAGES")) : 8;
    bool alternate = env_sw("HRN_ALTERNATE") ? atoi(env_sw("HRN_ALTERNATE")) != 0 : true;
    int block_order = env_sw("HRN_BLOCK_ORDER") ? atoi(env_sw("HRN_BLOCK_ORDER")) : 1;
    int long_factor = env_sw("HRN_LONG_FACTOR") ? atoi(env_sw("HRN_LONG_FACTOR")) : 4;
    double long_share = env_sw("HRN_LONG_SHARE") ? atof(env_sw("HRN_LONG_SHARE")) : 0.85;
    int head_slabs = 1, head_slab_px = 1024;
    static constexpr int kHeadSlabPxSmall = 256;
    int head_slabs_small = 1;
    // the head / decode split of a pass of nb crops: 1024-pixel slabs unless that leaves most of the chip idle
    int head_px_for(int nb) const { return (long)nb * head_slabs >= 192 ? head_slab_px : kHeadSlabPxSmall; }
    int head_slabs_for(int nb) const { return (long)nb * head_slabs >= 192 ? head_slabs : head_slabs_small; }
    float *part_val = nullptr;
    int *part_idx = nullptr;
    // crop pre-path scratch (grown on demand, never inside hrn_forward)
    unsigned char *pre_tmp = nullptr;
    size_t pre_tmp_bytes = 0;
    ResizeTaps *rs_taps = nullptr;   // single-person pre-path: tap tables of the last (frame size, interpolation), device
    int rs_taps_cap = 0;
    // One workspace per handle: passes on DIFFERENT streams must not overlap on the device.  Every entry point that runs a pass
    // records `pass_done` behind it; a call on another stream than the previous one waits for that event first (same stream:
    // ordered anyway, nothing is waited for).
    hipEvent_t pass_done = nullptr;
    hipStream_t pass_stream = nullptr;
    bool pass_enter(hipStream_t s) {
        if (pass_done && pass_stream != s) return hip_ok(hipStreamWaitEvent(s, pass_done, 0), "hipStreamWaitEvent");
        return true;
    }
    bool pass_leave(hipStream_t s) {
        if (!pass_done && !hip_ok(hipEventCreateWithFlags(&pass_done, hipEventDisableTiming), "hipEventCreate")) return false;
        pass_stream = s;
        return hip_ok(hipEventRecord(pass_done, s), "hipEventRecord");
    }
    hipEvent_t rs_done = nullptr;    // recorded behind the last resize launch: a call on ANOTHER stream rewrites the table after it
    hipStream_t rs_stream = nullptr;
    CropParams *pre_params = nullptr;
    int pre_params_cap = 0;
    // pinned host image of one call's crop parameters + boxes (the async uploads read it after the call returned);
    // kPreRing images in rotation, each rewritten only after the upload that last read it has completed
    static constexpr int kPreRing = 4;
    char *pre_pin[kPreRing] = {nullptr, nullptr, nullptr, nullptr};
    size_t pre_pin_bytes[kPreRing] = {0, 0, 0, 0};
    hipEvent_t pre_landed[kPreRing] = {nullptr, nullptr, nullptr, nullptr};
    unsigned pre_ring_next = 0;
    uint64_t map_clock = 0;     // LRU stamp of the block-map slots
    int64_t map_builds = 0;     // block maps built + uploaded since creation (hrn_map_rebuilds)
    float *tta_hm = nullptr;  // flip-TTA: heat-maps of the mirrored micro-batch (allocated on first use)
    int64_t workspace_bytes = 0;

#include "ctx_plan.inc"
#include "ctx_memory.inc"
#include "ctx_weights.inc"
#include "ctx_run.inc"
};

// ====================================================================================================
namespace {
// pass_enter() .. pass_leave() on EVERY exit: kernels already queued by a pass that failed half-way must be covered by
// the handle's "pass done" event too, or the next call on another stream could overlap them on the shared workspace
struct PassScope {
    hrn_ctx *h;
    hipStream_t s;
    bool entered = false, left = false;
    PassScope(hrn_ctx *h_, hipStream_t s_) : h(h_), s(s_) { entered = h->pass_enter(s); }
    bool leave() {
        left = true;
        return entered && h->pass_leave(s);
    }
    ~PassScope() {
        if (entered && !left) (void)h->pass_leave(s);
    }
};
}  // namespace

extern "C" {

const char *hrn_version(void) { return "hrnet_mi355 0.1 (gfx950)"; }

int hrn_create(hrn_handle *out, int c, int nof_joints, int height, int width, int dtype, int max_batch,
               int device_id) {
    return hrn_create_model(out, HRN_MODEL_HRNET, c, nof_joints, height, width, dtype, max_batch, device_id);
}

int hrn_create_model(hrn_handle *out, int model, int c, int nof_joints, int height, int width, int dtype, int max_batch,
                     int device_id) {
    if (!out) return 1;
    *out = nullptr;
    if (model != HRN_MODEL_HRNET && model != HRN_MODEL_POSERESNET) {
        g_create_error = "model must be HRN_MODEL_HRNET or HRN_MODEL_POSERESNET";
        return 2;
    }
    if (model == HRN_MODEL_POSERESNET && c != 50 && c != 101 && c != 152) {
        g_create_error = "PoseResNet: c is the ResNet size, 50 / 101 / 152 (the reference's 18 / 34 cannot run: modules.py:50)";
        return 2;
    }
    if (model == HRN_MODEL_HRNET && (c <= 0 || (c % 32 != 0 && c % 48 != 0))) {
        g_create_error = "c must be a positive multiple of 32 or of 48 (HRNet-W32 / W48 / W64 ...): the kernels tile output channels by 32 / 48 / 64";
        return 2;
    }
    if (height <= 0 || width <= 0 || height % 32 || width % 32) {
        g_create_error = "resolution must be a positive multiple of 32 in both dimensions";
        return 2;
    }
    if (nof_joints <= 0 || nof_joints > 32) {
        g_create_error = "nof_joints must be in [1, 32]";
        return 2;
    }
    if (dtype != HRN_F32 && dtype != HRN_BF16) {
        g_create_error = "dtype must be HRN_F32 or HRN_BF16";
        return 2;
    }
    if (max_batch <= 0) {
        g_create_error = "max_batch must be positive";
        return 2;
    }
    g_env_seen.clear();
    std::unique_ptr<hrn_ctx> h(new hrn_ctx());
    h->switches = g_env_seen;
    h->model = model;
    h->c = c, h->joints = nof_joints, h->H = height, h->W = width, h->dtype = dtype, h->max_batch = max_batch;
    h->device = device_id, h->plan_only = device_id < 0;
    h->esize = dtype == HRN_BF16 ? 2 : 4;
    if (!h->plan_only) {
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || device_id >= ndev) {
            g_create_error = "no such HIP device " + std::to_string(device_id) + " (" +
                             (e != hipSuccess ? hipGetErrorString(e) : "device count " + std::to_string(ndev)) + ")";
            return 3;
        }
    }
    h->build_plan();
    h->index_taps();
    // every conv launcher covers cout in tiles of 16*nr channels: a width that leaves a remainder would silently skip
    // channels (c = 80: 16 of branch 0's 80).  Multiples of 32 and of 48 never do.
    for (const ConvOp &cv : h->convs)
        if (cv.nr <= 0 || cv.cout % (16 * cv.nr) != 0) {
            g_create_error = "unsupported width: convolution '" + cv.conv + "' has " + std::to_string(cv.cout) +
                             " output channels, not a multiple of its " + std::to_string(16 * cv.nr) +
                             "-channel tile (use a width that is a multiple of 32 or of 48)";
            return 2;
        }
    if (!h->allocate()) {
        g_create_error = h->err.empty() ? "allocation failed" : h->err;
        h->free_all();
        return 4;
    }
    *out = h.release();
    return 0;
}

void hrn_destroy(hrn_handle h) {
    if (!h) return;
    h->free_all();
    delete h;
}

const char *hrn_last_error(hrn_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int hrn_load_weights(hrn_handle h, const hrn_tensor_desc *descs, int n) {
    if (!h || !descs) return 1;
    return h->load_weights(descs, n) ? 0 : 5;
}

int64_t hrn_weight_blob_bytes(hrn_handle h) { return h ? h->blob_bytes : 0; }
void *hrn_weight_blob_ptr(hrn_handle h) { return h ? h->blob : nullptr; }
int hrn_adopt_weights(hrn_handle h) {
    if (!h) return 1;
    h->weights_loaded = true;
    return 0;
}

int hrn_weight_blob_read(hrn_handle h, int64_t offset, void *dst, int64_t nbytes) {
    if (!h || !dst || offset < 0 || nbytes < 0 || offset + nbytes > h->blob_bytes) return 1;
    if (h->plan_only) {
        memcpy(dst, h->blob + offset, (size_t)nbytes);
        return 0;
    }
    if (!h->hip_ok(hipSetDevice(h->device), "hipSetDevice")) return 6;
    return h->hip_ok(hipMemcpy(dst, h->blob + offset, (size_t)nbytes, hipMemcpyDeviceToHost), "hipMemcpy") ? 0 : 6;
}

int hrn_forward(hrn_handle h, const void *images_dev, int n, const void *boxes_dev, int box_dtype, float *pts_dev,
                float *heatmaps_dev, void *stream) {
    if (!h) return 1;
    if (!h->check_forward_args(images_dev, n, boxes_dev, pts_dev, heatmaps_dev)) return 7;
    if (!h->hip_ok(hipSetDevice(h->device), "hipSetDevice")) return 6;
    const int hm = (h->H / 4) * (h->W / 4);
    if (n == 0) return 0;
    PassScope scope(h, (hipStream_t)stream);
    if (!scope.entered) return 6;
    for (int off = 0; off < n; off += h->max_batch) {
        const int nb = n - off < h->max_batch ? n - off : h->max_batch;
        const float *img = (const float *)images_dev + (size_t)off * 3 * h->H * h->W;
        const void *bx = boxes_dev ? (const char *)boxes_dev + (size_t)off * 16 : nullptr;
        float *p = pts_dev ? pts_dev + (size_t)off * h->joints * 3 : nullptr;
        float *hp = heatmaps_dev ? heatmaps_dev + (size_t)off * h->joints * hm : nullptr;
        if (!h->run_pass(img, nb, bx, box_dtype, p, hp, (hipStream_t)stream, nullptr)) return 8;
    }
    return scope.leave() ? 0 : 6;
}

// Flip test-time augmentation + the evaluation decode (testing/Test.py:132-140, training/COCO.py:206-230,
// misc/utils.py:9-29, 125-175): two passes per micro-batch (the second reads the crops mirrored in the stem), then one
// kernel averages, finds the maxima and applies the quarter-pixel refinement.
int hrn_forward_flip_tta(hrn_handle h, const void *images_dev, int n, const int32_t *flip_pairs_host, int npairs,
                         int post_processing, float *heatmaps_dev, float *preds_dev, float *maxvals_dev, void *stream) {
    if (!h) return 1;
    if (!h->check_forward_args(images_dev, n, nullptr, nullptr, heatmaps_dev, /*outputs_optional=*/true)) return 7;
    if (!heatmaps_dev || !preds_dev || !maxvals_dev || npairs < 0 || (npairs > 0 && !flip_pairs_host)) {
        h->err = "heatmaps, preds and maxvals are required outputs; flip_pairs must be npairs x 2";
        return 7;
    }
    TtaArgs a;
    for (int j = 0; j < 32; ++j) a.pair[j] = j;
    for (int k = 0; k < npairs; ++k) {
        const int p0 = flip_pairs_host[2 * k], p1 = flip_pairs_host[2 * k + 1];
        if (p0 < 0 || p1 < 0 || p0 >= h->joints || p1 >= h->joints) {
            h->err = "flip pair out of range";
            return 7;
        }
        // flip_back (misc/utils.py:24-27) swaps the two maps IN PLACE, pair after pair: compose the swaps in that order
        // (equal to "p0 <-> p1" only while no joint occurs in two pairs)
        std::swap(a.pair[p0], a.pair[p1]);
    }
    if (n == 0) return 0;
    if (!h->hip_ok(hipSetDevice(h->device), "hipSetDevice")) return 6;
    const int hh = h->H / 4, ww = h->W / 4, hm = hh * ww;
    if (!h->tta_hm &&
        !h->hip_ok(hipMalloc((void **)&h->tta_hm, (size_t)h->max_batch * h->joints * hm * sizeof(float)), "hipMalloc(flip-TTA)"))
        return 6;
    hipStream_t s = (hipStream_t)stream;
    PassScope scope(h, s);
    if (!scope.entered) return 6;
    for (int off = 0; off < n; off += h->max_batch) {
        const int nb = n - off < h->max_batch ? n - off : h->max_batch;
        const float *img = (const float *)images_dev + (size_t)off * 3 * h->H * h->W;
        float *out = heatmaps_dev + (size_t)off * h->joints * hm;
        if (!h->run_pass(img, nb, nullptr, 0, nullptr, out, s, nullptr, 0)) return 8;
        if (!h->run_pass(img, nb, nullptr, 0, nullptr, h->tta_hm, s, nullptr, 1)) return 8;
        a.hm = out, a.hm_flipped = h->tta_hm;
        a.preds = preds_dev + (size_t)off * h->joints * 2, a.maxvals = maxvals_dev + (size_t)off * h->joints;
        a.n = nb, a.joints = h->joints, a.h = hh, a.w = ww, a.post_processing = post_processing;
        if (!h->hip_ok(launch_tta_decode(a, s), "flip-TTA decode launch")) return 8;
    }
    return scope.leave() ? 0 : 6;
}

// SimpleHRNet.py:236-278.  The box arithmetic is Python's, restated in double: round() is round-half-even on a
// float, `//` on non-negative ints is C's `/`, int(round(x)) = nearbyint under the default rounding mode.
int hrn_preprocess_frame(hrn_handle h, const uint8_t *frame_dev, int frame_h, int frame_w, const float *dets_host,
                         int det_stride, int n, int variant, float *images_dev, int32_t *boxes_host, int32_t *boxes_dev,
                         void *stream) {
    if (!h) return 1;
    if (h->plan_only) {
        h->err = "plan-only handle (device_id < 0): there is no CPU compute path";
        return 7;
    }
    if (variant != HRN_CROP_PAD && variant != HRN_CROP_CLAMP) {
        h->err = "variant must be HRN_CROP_PAD or HRN_CROP_CLAMP";
        return 7;
    }
    if (n < 0 || det_stride < 4 || frame_h <= 0 || frame_w <= 0 || (n > 0 && (!frame_dev || !dets_host || !images_dev))) {
        h->err = "bad frame / detections / n";
        return 7;
    }
    if (n == 0) return 0;
    if (!h->hip_ok(hipSetDevice(h->device), "hipSetDevice")) return 6;
    const int H = h->H, W = h->W;
    // crop parameters + boxes are written straight into a pinned image the async uploads below read after this call
    // has returned; kPreRing images in rotation, each reused only once the upload that read it last has completed
    const size_t cp_bytes = ((size_t)n * sizeof(CropParams) + 63) / 64 * 64, need = cp_bytes + (size_t)n * 16;
    const unsigned ring = h->pre_ring_next++ % hrn_ctx::kPreRing;
    if (h->pre_landed[ring]) {
        if (!h->hip_ok(hipEventSynchronize(h->pre_landed[ring]), "hipEventSynchronize")) return 6;
    } else if (!h->hip_ok(hipEventCreateWithFlags(&h->pre_landed[ring], hipEventDisableTiming), "hipEventCreate")) {
        return 6;
    }
    if (need > h->pre_pin_bytes[ring]) {
        if (h->pre_pin[ring]) (void)hipHostFree(h->pre_pin[ring]);
        h->pre_pin[ring] = nullptr, h->pre_pin_bytes[ring] = 0;
        const size_t cap = std::max<size_t>(need * 2, 4096);
        if (!h->hip_ok(hipHostMalloc((void **)&h->pre_pin[ring], cap, hipHostMallocDefault), "hipHostMalloc(crop params)")) return 6;
        h->pre_pin_bytes[ring] = cap;
    }
    CropParams *cps = (CropParams *)h->pre_pin[ring];
    int32_t *boxes = (int32_t *)(h->pre_pin[ring] + cp_bytes);
    size_t tmp_bytes = 0;
    int max_h_pad = 0;
    for (int i = 0; i < n; ++i) {
        const float *d = dets_host + (size_t)i * det_stride;
        const long x1 = (long)std::nearbyint((double)d[0]), y1 = (long)std::nearbyint((double)d[1]);
        const long x2 = (long)std::nearbyint((double)d[2]), y2 = (long)std::nearbyint((double)d[3]);
        if (x2 <= x1 || y2 <= y1) {
            h->err = "detection " + std::to_string(i) + " is degenerate";
            return 7;
        }
        const double cf = (double)H / (double)W * (double)(x2 - x1) / (double)(y2 - y1);
        // The reference slices numpy arrays with these numbers: a negative start would wrap around.  The PAD variant
        // slices with the rounded box itself; the CLAMP variant re-derives (and clamps to the frame) the side it
        // enlarges, so only the OTHER side has to be inside the frame as given (SimpleHRNet.py:396-407).
        const bool x_as_given = variant == HRN_CROP_PAD || !(cf < 1), y_as_given = variant == HRN_CROP_PAD || !(cf > 1);
        if ((x_as_given && (x1 < 0 || x1 >= frame_w)) || (y_as_given && (y1 < 0 || y1 >= frame_h))) {
            h->err = "detection " + std::to_string(i) + " starts outside the frame";
            return 7;
        }
        long x1n = x1, x2n = x2, y1n = y1, y2n = y2, pt = 0, pb = 0, pl = 0, pr = 0;
        long sx1 = x1, sy1 = y1, sx2 = x2, sy2 = y2;  // what is sliced out of the frame
        if (variant == HRN_CROP_CLAMP) {  // SimpleHRNet.py:396-407: enlarge, clamp to the frame, slice the enlarged box
            if (cf > 1) {
                const long center = y1 + (y2 - y1) / 2;
                const long length = (long)std::nearbyint((double)(y2 - y1) * cf);
                y1n = std::max<long>(0, center - length / 2), y2n = std::min<long>(frame_h, center + length / 2);
            } else if (cf < 1) {
                const long center = x1 + (x2 - x1) / 2;
                const long length = (long)std::nearbyint((double)(x2 - x1) * 1 / cf);
                x1n = std::max<long>(0, center - length / 2), x2n = std::min<long>(frame_w, center + length / 2);
            }
            sx1 = x1n, sy1 = y1n, sx2 = x2n, sy2 = y2n;
            if (sx2 <= sx1 || sy2 <= sy1 || sx1 >= frame_w || sy1 >= frame_h) {
                h->err = "detection " + std::to_string(i) + " is degenerate after clamping";
                return 7;
            }
        } else if (cf > 1) {  // increase y side
            const long center = y1 + (y2 - y1) / 2;
            const long length = (long)std::nearbyint((double)(y2 - y1) * cf);
            y1n = center - length / 2, y2n = center + length / 2;
            pt = std::labs(y1n - y1), pb = std::labs(y2n - y2);
        } else if (cf < 1) {
            const long center = x1 + (x2 - x1) / 2;
            const long length = (long)std::nearbyint((double)(x2 - x1) * 1 / cf);
            x1n = center - length / 2, x2n = center + length / 2;
            pl = std::labs(x1n - x1), pr = std::labs(x2n - x2);
        }
        CropParams &cp = cps[i];
        cp.x1 = (int)sx1, cp.y1 = (int)sy1;
        cp.w_crop = (int)(std::min<long>(sx2, frame_w) - sx1), cp.h_crop = (int)(std::min<long>(sy2, frame_h) - sy1);  // numpy slicing
        cp.pad_top = (int)pt, cp.pad_left = (int)pl;
        cp.h_pad = cp.h_crop + (int)(pt + pb), cp.w_pad = cp.w_crop + (int)(pl + pr);
        cp.tmp_off = (long long)tmp_bytes;
        tmp_bytes += ((size_t)cp.h_pad * W * 3 + 255) / 256 * 256;
        if (cp.h_pad > max_h_pad) max_h_pad = cp.h_pad;
        boxes[(size_t)i * 4 + 0] = (int32_t)x1n, boxes[(size_t)i * 4 + 1] = (int32_t)y1n;
        boxes[(size_t)i * 4 + 2] = (int32_t)x2n, boxes[(size_t)i * 4 + 3] = (int32_t)y2n;
    }
    hipStream_t s = (hipStream_t)stream;
    if (tmp_bytes > h->pre_tmp_bytes || n > h->pre_params_cap) {  // grow the scratch: wait for whoever still reads the old one
        if (!h->hip_ok(hipStreamSynchronize(s), "hipStreamSynchronize")) return 6;
        if (tmp_bytes > h->pre_tmp_bytes) {
            if (h->pre_tmp) (void)hipFree(h->pre_tmp);
            h->pre_tmp = nullptr, h->pre_tmp_bytes = 0;
            if (!h->hip_ok(hipMalloc((void **)&h->pre_tmp, tmp_bytes), "hipMalloc(pre-path scratch)")) return 6;
            h->pre_tmp_bytes = tmp_bytes;
        }
        if (n > h->pre_params_cap) {
            if (h->pre_params) (void)hipFree(h->pre_params);
            h->pre_params = nullptr, h->pre_params_cap = 0;
            if (!h->hip_ok(hipMalloc((void **)&h->pre_params, (size_t)n * sizeof(CropParams)), "hipMalloc(crop params)")) return 6;
            h->pre_params_cap = n;
        }
    }
    if (!h->hip_ok(hipMemcpyAsync(h->pre_params, cps, (size_t)n * sizeof(CropParams), hipMemcpyHostToDevice, s),
                   "hipMemcpyAsync(crop params)"))
        return 6;
    if (boxes_dev && !h->hip_ok(hipMemcpyAsync(boxes_dev, boxes, (size_t)n * 16, hipMemcpyHostToDevice, s),
                                "hipMemcpyAsync(boxes)"))
        return 6;
    if (!h->hip_ok(hipEventRecord(h->pre_landed[ring], s), "hipEventRecord")) return 6;
    if (boxes_host) memcpy(boxes_host, boxes, (size_t)n * 16);
    if (!h->hip_ok(launch_prepath(frame_dev, frame_w, h->pre_params, n, max_h_pad, h->pre_tmp, images_dev, H, W, s),
                   "pre-path launch"))
        return 8;
    return 0;
}

int hrn_resize_frames(hrn_handle h, const uint8_t *frames_dev, int n, int frame_h, int frame_w, int interpolation,
                      float *images_dev, void *stream) {
    if (!h) return 1;
    if (h->plan_only) {
        h->err = "plan-only handle (device_id < 0): there is no CPU compute path";
        return 7;
    }
    if (interpolation != HRN_INTER_NEAREST && interpolation != HRN_INTER_LINEAR && interpolation != HRN_INTER_CUBIC) {
        h->err = "interpolation must be HRN_INTER_NEAREST (0), HRN_INTER_LINEAR (1) or HRN_INTER_CUBIC (2)";
        return 7;
    }
    if (n < 0 || frame_h <= 0 || frame_w <= 0 || (n > 0 && (!frames_dev || !images_dev))) {
        h->err = "bad frames / n";
        return 7;
    }
    if (n == 0) return 0;
    if (!h->hip_ok(hipSetDevice(h->device), "hipSetDevice")) return 6;
    hipStream_t s = (hipStream_t)stream;
    const int H = h->H, W = h->W;
    if (W + H > h->rs_taps_cap) {
        if (h->rs_taps) {
            if (!h->hip_ok(hipStreamSynchronize(s), "hipStreamSynchronize")) return 6;
            (void)hipFree(h->rs_taps);
            h->rs_taps = nullptr, h->rs_taps_cap = 0;
        }
        if (!h->hip_ok(hipMalloc((void **)&h->rs_taps, (size_t)(W + H) * sizeof(ResizeTaps)), "hipMalloc(resize taps)")) return 6;
        h->rs_taps_cap = W + H;
    }
    // the one tap table of the handle is rewritten by every call: a call on another stream than the previous one waits until
    // that one's kernels have read it (same stream: ordered anyway)
    if (h->rs_done && h->rs_stream != s && !h->hip_ok(hipStreamWaitEvent(s, h->rs_done, 0), "hipStreamWaitEvent")) return 6;
    if (!h->rs_done && !h->hip_ok(hipEventCreateWithFlags(&h->rs_done, hipEventDisableTiming), "hipEventCreate")) return 6;
    if (!h->hip_ok(launch_resize_frames(frames_dev, n, frame_h, frame_w, interpolation, h->rs_taps, images_dev, H, W, s), "resize launch"))
        return 8;
    h->rs_stream = s;
    if (!h->hip_ok(hipEventRecord(h->rs_done, s), "hipEventRecord")) return 6;
    return 0;
}

// Debug tap: one micro-batch with the named tensor copied out right after the launch that completes it.
int hrn_tap_count(hrn_handle h) { return h ? (int)h->taps.size() : 0; }

int hrn_get_tap_info(hrn_handle h, int index, hrn_tap_info *out) {
    if (!h || !out || index < 0 || index >= (int)h->taps.size()) return 1;
    const TapPoint &tp = h->taps[index];
    const Tensor &t = h->tensors[tp.tensor];
    memset(out, 0, sizeof *out);
    snprintf(out->name, sizeof out->name, "%s", tp.name.c_str());
    out->c = t.c, out->h = t.h, out->w = t.w, out->conv_index = tp.conv;
    return 0;
}

int hrn_forward_tap(hrn_handle h, const void *images_dev, int n, const char *tap_name, int crop0, int ncrops, int crop_step,
                    float *dst_dev, float *heatmaps_dev, void *stream) {
    if (!h) return 1;
    if (!h->check_forward_args(images_dev, n, nullptr, nullptr, nullptr, /*outputs_optional=*/true)) return 7;
    if (!tap_name || !dst_dev || n < 1 || n > h->max_batch || crop0 < 0 || ncrops < 1 || crop_step < 1 ||
        (int64_t)crop0 + (int64_t)(ncrops - 1) * crop_step >= n) {
        h->err = "hrn_forward_tap: needs a tap name, a destination, 1 <= n <= max_batch and crops crop0, crop0 + step, ... inside the call";
        return 7;
    }
    const TapPoint *tp = nullptr;
    for (const TapPoint &t : h->taps)
        if (t.name == tap_name) tp = &t;
    if (!tp) {
        h->err = std::string("hrn_forward_tap: no tensor named '") + tap_name + "' is written by this plan";
        return 7;
    }
    if (tp->conv >= 0 && h->fused_now(h->convs[tp->conv], n)) {
        h->err = std::string("hrn_forward_tap: '") + tap_name + "' stays in LDS at this batch size (fused BasicBlock pass); tap the block's conv2";
        return 7;
    }
    if (!h->hip_ok(hipSetDevice(h->device), "hipSetDevice")) return 6;
    const TapReq req{tp->op, tp->tensor, crop0, ncrops, crop_step, dst_dev};
    PassScope scope(h, (hipStream_t)stream);
    if (!scope.entered) return 6;
    if (!h->run_pass((const float *)images_dev, n, nullptr, 0, nullptr, heatmaps_dev, (hipStream_t)stream, nullptr, 0, &req)) return 8;
    return scope.leave() ? 0 : 6;
}

int hrn_conv_count(hrn_handle h) { return h ? (int)h->convs.size() : 0; }

int hrn_get_conv_info(hrn_handle h, int index, hrn_conv_info *out) {
    if (!h || !out || index < 0 || index >= (int)h->convs.size()) return 1;
    const ConvOp &cv = h->convs[index];
    memset(out, 0, sizeof *out);
    snprintf(out->name, sizeof out->name, "%s", cv.conv.c_str());
    out->cin = cv.cin, out->cout = cv.cout, out->ksize = cv.k, out->stride = cv.stride, out->relu = cv.relu;
    out->has_residual = cv.res_t >= 0;
    out->in_h = h->tensors[cv.in_t].h, out->in_w = h->tensors[cv.in_t].w;
    out->out_h = h->tensors[cv.out_t].h, out->out_w = h->tensors[cv.out_t].w;
    out->kpad = cv.kpad, out->nr = cv.nr, out->algo = cv.fuse_with >= 0 || cv.fused_away ? 2 : cv.n96 ? 3 : cv.s2 ? 4 : cv.algo, out->ks = cv.ks;
    out->w_offset = cv.w_off, out->w_bytes = cv.w_bytes, out->b_offset = cv.b_off;
    out->flops = cv.flops;
    return 0;
}

double hrn_flops_per_crop(hrn_handle h) {
    if (!h) return 0;
    double f = 2.0 * 64 * (h->model == 1 ? 147 : 27) * (h->H / 2) * (double)(h->W / 2);   // conv1
    for (auto &cv : h->convs) f += cv.flops;
    f += 2.0 * h->joints * h->head_c * (h->H / 4) * (double)(h->W / 4);      // final_layer
    return f;
}

int64_t hrn_workspace_bytes(hrn_handle h) { return h ? h->workspace_bytes : 0; }
int64_t hrn_map_rebuilds(hrn_handle h) { return h ? h->map_builds : -1; }
int hrn_launches_per_pass(hrn_handle h) { return h ? (int)h->ops.size() - (h->stem_fuse ? 1 : 0) : 0; }
int hrn_stem_fused(hrn_handle h) { return h && h->stem_fuse ? 1 : 0; }
int hrn_conv_compact(hrn_handle h, int index) { return h && index >= 0 && index < (int)h->convs.size() && h->convs[index].compact ? 1 : 0; }
const char *hrn_switches(hrn_handle h) { return h ? h->switches.c_str() : ""; }

int64_t hrn_debug_pad_violations(hrn_handle h) {
    if (!h || h->plan_only) return -1;
    if (!h->hip_ok(hipSetDevice(h->device), "hipSetDevice")) return -1;
    unsigned long long *cnt = nullptr, host = 0;
    if (!h->hip_ok(hipMalloc((void **)&cnt, sizeof *cnt), "hipMalloc") || !h->hip_ok(hipMemset(cnt, 0, sizeof *cnt), "hipMemset")) return -1;
    bool ok = h->hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize");
    for (const Buffer &b : h->buffers) {
        if (!ok) break;
        PadCheckArgs a;
        a.buf = b.dev, a.count = cnt, a.rows = (long)b.rows, a.lead_rows = (long)b.lead_rows;
        a.c = b.c, a.h = b.h, a.w = b.w, a.wp = b.w + 1, a.hpwp = (b.h + 1) * (b.w + 1), a.nmax = h->max_batch;
        ok = h->hip_ok(launch_pad_check(h->dtype, a, nullptr), "pad check launch");
    }
    ok = ok && h->hip_ok(hipMemcpy(&host, cnt, sizeof host, hipMemcpyDeviceToHost), "hipMemcpy");
    (void)hipFree(cnt);
    return ok ? (int64_t)host : -1;
}

int hrn_plan_block_map(hrn_handle h, int group, int n, int reverse, int32_t *blocks, int capacity, int32_t *members, int member_capacity) {
    if (!h || n <= 0 || n > h->max_batch) return -1;
    if (group < 0 || group >= (int)h->groups.size()) return -1;
    const Conv3Group &g = h->groups[group];
    std::vector<int2> map;
    std::vector<int> px;
    const int nblocks = h->group_blocks(g, n, &map, reverse != 0, false, &px);
    for (int i = 0; i < nblocks && i < capacity; ++i) {
        const int x = map[i].x, y = map[i].y;
        int32_t *o = blocks + (size_t)i * 6;
        o[0] = x & 0xff, o[1] = (x >> 8) & 0xff, o[2] = x >> 16, o[3] = y & 0x1fffffff, o[4] = px[i],
        o[5] = ((y >> 29) & 1) | (((y >> 30) & 1) << 1);
    }
    int nm = 0;  // descriptor -> convolution: the plain ones first, then the fused forms (conv1's index + 2^30)
    for (size_t k = 0; k < g.conv_idx.size(); ++k, ++nm)
        if (nm < member_capacity) members[nm] = g.conv_idx[k];
    for (size_t k = 0; k < g.conv_idx.size(); ++k)
        if (g.fused_prob[k] >= 0) {
            if (g.fused_prob[k] < member_capacity) members[g.fused_prob[k]] = g.conv_idx[k] | (1 << 30);
            ++nm;
        }
    return nblocks;
}

int hrn_plan_direct_map(hrn_handle h, int group, int n, int32_t *blocks, int capacity, int32_t *members, int member_capacity,
                        int32_t *pixels_per_tile) {
    if (!h || n <= 0 || n > h->max_batch) return -1;
    if (group < 0 || group >= (int)h->dgroups.size()) return -1;
    const DirectGroup &g = h->dgroups[group];
    int mr = 4;  // as run_pass chooses it
    while (mr > 1 && h->direct_group_blocks(g, n, nullptr, mr) < 512) mr >>= 1;
    std::vector<int2> map;
    const int nblocks = h->direct_group_blocks(g, n, &map, mr);
    for (int i = 0; i < nblocks && i < capacity; ++i)
        blocks[(size_t)i * 3] = map[i].x & 0xff, blocks[(size_t)i * 3 + 1] = map[i].x >> 8, blocks[(size_t)i * 3 + 2] = map[i].y;
    for (size_t k = 0; k < g.conv_idx.size() && (int)k < member_capacity; ++k) members[k] = g.conv_idx[k];
    if (pixels_per_tile) *pixels_per_tile = 64 * mr;
    return nblocks;
}

int hrn_plan_s2_map(hrn_handle h, int group, int n, int32_t *blocks, int capacity, int32_t *parts, int part_capacity, int32_t *active) {
    if (!h || n <= 0 || n > h->max_batch) return -1;
    // group -1: the fused stem's problem (conv2 with one output row per tile), when the handle runs it (hrn_stem_fused)
    if (group == -1 && !h->stem_fuse) return -1;
    if (group < -1 || group >= (int)h->s2groups.size()) return -1;
    const S2Group &g = group == -1 ? h->stemf : h->s2groups[group];
    std::vector<int2> map;
    const int nblocks = h->s2_blocks(g, n, &map);
    for (int i = 0; i < nblocks && i < capacity; ++i)
        blocks[(size_t)i * 3] = map[i].x & 0xff, blocks[(size_t)i * 3 + 1] = map[i].x >> 8, blocks[(size_t)i * 3 + 2] = map[i].y;
    int np = 0;  // per part: problem, convolution, 48-cout tile, rows per tile, tiles per image
    for (size_t k = 0; k < g.probs.size(); ++k)
        for (auto &pt : g.probs[k].parts) {
            if (np < part_capacity) {
                int32_t *o = parts + (size_t)np * 5;
                o[0] = (int32_t)k, o[1] = pt.first, o[2] = pt.second, o[3] = g.probs[k].rows, o[4] = g.probs[k].tiles_per_image;
            }
            ++np;
        }
    if (active) *active = h->s2_active(g, n) ? 1 : 0;
    return nblocks | (np << 20);
}

int hrn_profile_pass(hrn_handle h, const void *images_dev, int n, float *conv_ms, int conv_ms_len, float *other_ms,
                     void *stream) {
    if (!h) return 1;
    if (n > h->max_batch) n = h->max_batch;
    // profiling computes heat-map partials only (no pts): boxes are not needed
    if (!h->check_forward_args(images_dev, n, nullptr, nullptr, nullptr, /*outputs_optional=*/true)) return 7;
    if (!h->hip_ok(hipSetDevice(h->device), "hipSetDevice")) return 6;
    hrn_ctx::Timing tm;
    tm.ev.resize(h->ops.size() + 1);
    for (auto &e : tm.ev)
        if (!h->hip_ok(hipEventCreate(&e), "hipEventCreate")) return 6;
    PassScope scope(h, (hipStream_t)stream);
    bool ok = scope.entered && h->run_pass((const float *)images_dev, n, nullptr, 0, nullptr, nullptr, (hipStream_t)stream, &tm) && scope.leave();
    ok = ok && h->hip_ok(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize");
    if (ok) {
        if (other_ms) other_ms[0] = other_ms[1] = other_ms[2] = other_ms[3] = 0.f;
        for (int i = 0; i < conv_ms_len; ++i) conv_ms[i] = 0.f;
        for (size_t oi = 0; oi < h->ops.size(); ++oi) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, tm.ev[oi], tm.ev[oi + 1]);
            const Op &op = h->ops[oi];
            if (op.kind == OP_CONV) {
                if (conv_ms && op.idx < conv_ms_len) conv_ms[op.idx] = ms;
            } else if (op.kind == OP_CONV3_GROUP) {  // one launch, several convs: split by FLOPs
                const Conv3Group &g = h->groups[op.idx];
                double tot = 0;  // (a fused BasicBlock's launch carries its conv2 as well, and conv2's launch skips it)
                for (int ci : g.conv_idx) {
                    const ConvOp &cv = h->convs[ci];
                    if (h->skipped(cv, n)) continue;
                    tot += cv.flops + (h->fused_now(cv, n) ? h->convs[cv.fuse_with].flops : 0.0);
                }
                for (int ci : g.conv_idx) {
                    const ConvOp &cv = h->convs[ci];
                    if (h->skipped(cv, n)) continue;
                    if (conv_ms && ci < conv_ms_len) conv_ms[ci] = (float)(ms * cv.flops / tot);
                    const int c2 = h->fused_now(cv, n) ? cv.fuse_with : -1;
                    if (conv_ms && c2 >= 0 && c2 < conv_ms_len) conv_ms[c2] = (float)(ms * h->convs[c2].flops / tot);
                }
            } else if (op.kind == OP_CONV_GROUP) {  // likewise; these are latency / bandwidth bound: split by block count
                const DirectGroup &g = h->dgroups[op.idx];
                double tot = 0;
                std::vector<double> wgt;
                for (int ci : g.conv_idx) {
                    const ConvOp &cv = h->convs[ci];
                    wgt.push_back((double)h->grid_of(cv).hpwp * (cv.cout / (16 * g.nr)) * cv.kchunks);
                    tot += wgt.back();
                }
                for (size_t k = 0; k < g.conv_idx.size(); ++k)
                    if (conv_ms && g.conv_idx[k] < conv_ms_len) conv_ms[g.conv_idx[k]] = (float)(ms * wgt[k] / tot);
            } else if (op.kind == OP_S2_GROUP) {  // one launch (or its generic fallback): split by FLOPs
                const S2Group &g = h->s2groups[op.idx];
                double tot = 0;
                for (int ci : g.conv_idx) tot += h->convs[ci].flops;
                for (int ci : g.conv_idx)
                    if (conv_ms && ci < conv_ms_len) conv_ms[ci] = (float)(ms * h->convs[ci].flops / tot);
            } else if (op.kind == OP_CHAIN) {  // 1x1 convs of equal FLOPs (+ the 3x3 in front of them): split by FLOPs
                const hrn_ctx::Chain &ch = h->chains[op.idx];
                double tot = 0;
                for (int ci : {ch.conv3, ch.conv1, ch.ds, ch.conv2})
                    if (ci >= 0) tot += h->convs[ci].flops;
                for (int ci : {ch.conv3, ch.conv1, ch.ds, ch.conv2})
                    if (conv_ms && ci >= 0 && ci < conv_ms_len) conv_ms[ci] = (float)(ms * h->convs[ci].flops / tot);
            } else if (other_ms) {
                const int slot = (op.kind == OP_STEM || op.kind == OP_STEM7 || op.kind == OP_MAXPOOL) ? 0 : op.kind == OP_FUSE ? 1 : op.kind == OP_HEAD ? 2 : 3;
                other_ms[slot] += ms;
            }
        }
    }
    for (auto &e : tm.ev) (void)hipEventDestroy(e);
    return ok ? 0 : 8;
}

}  // extern "C"
