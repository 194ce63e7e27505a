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

using namespace hrn;

std::string g_create_error;

struct Tensor {
    int c = 0, h = 0, w = 0;
    int wp = 0, hp = 0, hpwp = 0;
    int buf = -1;
};

struct Buffer {
    int c, h, w;
    size_t lead_rows, rows, bytes;
    char *dev = nullptr;  // allocation start
    bool in_use = false;
};

enum OpKind { OP_STEM, OP_STEM7, OP_MAXPOOL, OP_CONV, OP_CONV3_GROUP, OP_CONV_GROUP, OP_S2_GROUP, OP_CHAIN, OP_FUSE, OP_HEAD, OP_DECODE };

struct ConvOp {
    std::string conv, bn;  // state_dict prefixes ("" bn => plain bias conv)
    int in_t, out_t, res_t;
    int cin, cout, k, stride, relu;
    int kpad, kchunks, nr;
    int algo = 0;          // 0 = generic kernel (kernels.hip), 1 = pipelined LDS-staged 3x3 stride 1 (conv3x3_lds.hip)
    int up = 0;            // 1 + 2a + b: phase (a, b) of a ConvTranspose2d(4, s2, p1) as a 3x3 conv on the input grid
    int ks = 0, slices = 0, ntiles = 0, nch = 0;
    bool n96 = false;      // 96-cout form of the LDS-staged kernel (conv3x3_n96.inc): ks = 32, nr = 6, same launch family as (48, 3)
    int fuse_with = -1;    // conv1 of a BasicBlock that can also compute this conv2 (conv3x3_lds.hip: bbf_run)
    bool fused_away = false;  // conv2 of such a block: skipped in its own launch whenever conv1's launch ran fused
    int64_t w_off = 0, w_bytes = 0, b_off = 0;
    // stride-2 slab kernel (conv_s2.hip): 3x3 / stride 2 / 48 input channels in bf16.  Such a convolution keeps its generic
    // plan (the small-call fallback, bit-identical) and carries a second weight image, the (48, 3) LDS form, for the slab kernel
    bool s2 = false;
    int64_t w2_off = 0, w2_bytes = 0;
    double flops = 0;
};

// Block maps (and, for the generic kernel, the descriptors) depend on the micro-batch size nb.  A call whose n is not a
// multiple of max_batch alternates two sizes, a serving loop a few more: every grouped launch keeps kMapSlots device
// copies keyed by nb (least recently used one replaced), each with its own PINNED host image and an event, so that
// (a) a steady mix of sizes uploads nothing, (b) the async H2D never reads pageable or short-lived memory, (c) a pinned
// image is only rewritten once its previous upload has completed.
constexpr int kMapSlots = 4;
struct MapSlot {
    int nb = -1;
    int nblocks = 0;
    int mr = 4;                  // generic kernel: 16-pixel fragments per wave of this map
    int2 *dev = nullptr, *pin = nullptr;
    ConvArgs *args_dev = nullptr, *args_pin = nullptr;  // generic kernel only
    // persistent work-queue form of a grouped BasicBlock launch (conv3x3_queue.inc): unit records instead of a block map
    QUnit *q_dev = nullptr, *q_pin = nullptr;
    int q_units = -1;            // -1: this size takes the per-block form
    int q_bbf_prob = 0, q_bbf_blocks = 0, q_bbf_tiles = 0;
    hipEvent_t landed = nullptr;
    hipStream_t up_stream = nullptr;   // the stream the upload went out on: a hit from ANOTHER stream waits for `landed` first
    uint64_t stamp = 0;
};

// a set of independent LDS-staged 3x3 convolutions issued as ONE launch (conv3x3_lds.hip)
struct Conv3Group {
    std::vector<int> conv_idx;
    std::vector<int> fused_prob;  // per member: index (within the group) of its fused-BasicBlock descriptor, or -1
    int prob_first = 0;          // index of the group's first descriptor in the device array
    int max_wp = 0;
    int64_t map_capacity = 0;    // blocks at max_batch
    MapSlot slot[kMapSlots];
    std::vector<int2> map_host;  // scratch of group_blocks()
    std::vector<QUnit> units_host;   // scratch of queue_plan()
};

// a set of independent convolutions on the generic kernel issued as ONE launch (kernels.hip: conv_direct_group_kernel)
struct DirectGroup {
    std::vector<int> conv_idx;
    int nr = 0;
    int64_t map_capacity = 0;
    MapSlot slot[kMapSlots];
    std::vector<int2> map_host;  // scratch of direct_group_blocks()
};

struct Op {
    OpKind kind;
    int idx;  // index into convs / fuses
};

// a set of stride-2 convolutions with 48 input channels issued as ONE launch of the slab kernel (conv_s2.hip): one problem
// per (input tensor, up to 8 parts of 48 output channels); `fallback` = the same convolutions on the generic kernel, taken
// when the call has too few tiles to fill the chip (same K order and arithmetic: bit-identical)
struct S2Group {
    struct Prob {
        int in_t;
        std::vector<std::pair<int, int>> parts;  // (convolution, 48-cout tile of it)
        int rows = 1, tiles_per_image = 1;
    };
    std::vector<int> conv_idx;
    std::vector<Prob> probs;
    std::vector<Op> fallback;
    int prob_first = 0;
    int64_t map_capacity = 0;
    MapSlot slot[kMapSlots];
    std::vector<int2> map_host;
};

struct FuseOp {
    int term_t[4];
    int shift[4];
    int nterms;
    int out_t;
    std::string name;  // "<stage>.fuse.<i>": the i-th output of the module's fuse (debug tap)
    int group = 1;     // this many consecutive fuses, starting here, go out as ONE launch (0: a member launched by its leader)
};

// Debug tap (hrn_forward_tap): a tensor some launch of the pass writes to HBM, by name
struct TapPoint {
    std::string name;
    int tensor;     // index into tensors
    int op;         // the tensor is complete after ops[op]
    int conv;       // convolution that writes it (-1: stem / fuse)
};
struct TapReq {
    int op, tensor, crop0, ncrops, crop_step;
    float *dst;
};

// x / d == (x * magic) >> shift for 0 <= x < 2^27
inline void fast_div(int d, unsigned *magic, int *shift) {
    int l = 0;
    while ((1 << l) < d) ++l;
    *shift = 30 + l;
    *magic = (unsigned)((1ull << *shift) / (unsigned)d + 1);
}

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

inline uint16_t f32_to_bf16_host(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

}  // namespace

struct hrn_ctx {
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
    S2Problem *s2probs_dev = nullptr;
    struct Chain {
        int conv3, conv1, ds;  // conv3 of Bottleneck b, conv1 of Bottleneck b+1, projection shortcut folded in (or -1)
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

    bool disable_lds = getenv("HRN_DISABLE_LDS") != nullptr;
    bool disable_group = getenv("HRN_DISABLE_GROUP") != nullptr;
    bool disable_dgroup = getenv("HRN_DISABLE_DGROUP") != nullptr;
    bool disable_chain = getenv("HRN_DISABLE_CHAIN") != nullptr;
    bool small_tiles = getenv("HRN_SMALL_TILES") ? atoi(getenv("HRN_SMALL_TILES")) != 0 : true;
    int small_below = getenv("HRN_SMALL_BELOW") ? atoi(getenv("HRN_SMALL_BELOW")) : 384;
    bool disable_chain_ds = getenv("HRN_DISABLE_CHAIN_DS") != nullptr;
    // generic conv kernel, bf16: the block's weights through LDS instead of one copy per wave from L2 (+1.6 % on the pass)
    bool direct_wlds = !(getenv("HRN_DIRECT_WLDS") && atoi(getenv("HRN_DIRECT_WLDS")) == 0);
    // fused BasicBlocks on the 48-channel branch (conv3x3_lds.hip: bbf_run): bit-identical, 2.5x less HBM traffic on that
    // branch, +2.6 % on the whole pass at 256 crops; HRN_BBF=0 goes back to two launches per block
    bool disable_bbf = getenv("HRN_BBF") && atoi(getenv("HRN_BBF")) == 0;
    int bbf_tpb_div = getenv("HRN_BBF_TPB_DIV") ? std::max(1, atoi(getenv("HRN_BBF_TPB_DIV"))) : 3;  // a fused tile ~ 3 plain ones
    // fused only when the call has at least this many 512-pixel tiles (four per CU): measured at 384x288 +2 % at 256
    // crops (3541 tiles), +1 % at 128-192, +6 % at 96 (1328 tiles), -1 % at 64 (885 tiles), -2 % at 20, -5 % at one crop,
    // where the two plain launches with their smaller tiles spread the work over more CUs
    int bbf_min_tiles = getenv("HRN_BBF_MIN_TILES") ? atoi(getenv("HRN_BBF_MIN_TILES")) : 1100;
    // 1: convolutions that read the same tensor share one cout-tile width so that they can share a launch (and L2)
    int dgroup_nr_mode = getenv("HRN_DGROUP_NR") ? atoi(getenv("HRN_DGROUP_NR")) : 1;
    bool disable_lds32 = getenv("HRN_DISABLE_LDS32") != nullptr;
    bool disable_n96 = getenv("HRN_DISABLE_N96") != nullptr;
    bool disable_fgroup = getenv("HRN_DISABLE_FGROUP") != nullptr;   // one launch per fuse output instead of one per StageModule
    bool disable_f32lds = getenv("HRN_DISABLE_F32LDS") != nullptr;   // fp32 3x3 stride-1 convs back on the generic kernel
    int f32_small_slices = getenv("HRN_F32_SMALL_SLICES") ? atoi(getenv("HRN_F32_SMALL_SLICES")) : 0;   // fp32: 128-pixel tiles from this many slices on (0: never; 4 and 8 measured: no gain)
    // stride-2 slab kernel (conv_s2.hip) off: those convolutions stay on the generic kernel (bit-identical results)
    // persistent work-queue form of the grouped BasicBlock launches (round 4; bit-identical to the per-block form and, measured
    // in the net, no faster: profiles/EXPERIMENTS.md -- so it is an option, HRN_QUEUE=1, not the default); tiles per unit; scale of
    // the share of the CUs that start on the fused 48-channel range; fewest units per CU for a launch to take the form
    bool queue_on = getenv("HRN_QUEUE") && atoi(getenv("HRN_QUEUE")) != 0;
    int queue_tpb = getenv("HRN_Q_TPB") ? std::max(1, atoi(getenv("HRN_Q_TPB"))) : 1;
    double queue_bbf_scale = getenv("HRN_Q_BBF_SCALE") ? atof(getenv("HRN_Q_BBF_SCALE")) : 1.2;
    int queue_min_units_per_cu = getenv("HRN_Q_MIN_UNITS") ? atoi(getenv("HRN_Q_MIN_UNITS")) : 2;
    int num_cus = 256;
    int *qheads_dev = nullptr;   // 16 ints per grouped launch (8 used: one list head per XCD), zeroed at the start of every pass
    std::vector<Conv3Problem> probs_host;
    bool disable_s2 = getenv("HRN_DISABLE_S2") != nullptr;
    // the slab kernel is taken when a launch has at least this many tiles (one per CU); smaller calls use the generic kernel
    int s2_min_tiles = getenv("HRN_S2_MIN_TILES") ? atoi(getenv("HRN_S2_MIN_TILES")) : 256;
    int s2_target_blocks = getenv("HRN_S2_BLOCKS") ? std::max(1, atoi(getenv("HRN_S2_BLOCKS"))) : 256;   // 96-cout form off: those convolutions take the (48, 3) form
    bool disable_stem_mfma = getenv("HRN_DISABLE_STEM_MFMA") != nullptr;
    bool disable_head_mfma = getenv("HRN_DISABLE_HEAD_MFMA") != nullptr;
    bool direct_nr6 = getenv("HRN_DIRECT_NR6") ? atoi(getenv("HRN_DIRECT_NR6")) != 0 : true;
    int half_stages_per_block = getenv("HRN_HALF_STAGES") ? atoi(getenv("HRN_HALF_STAGES")) : 8;
    bool alternate = getenv("HRN_ALTERNATE") ? atoi(getenv("HRN_ALTERNATE")) != 0 : true;
    int block_order = getenv("HRN_BLOCK_ORDER") ? atoi(getenv("HRN_BLOCK_ORDER")) : 1;
    int long_factor = getenv("HRN_LONG_FACTOR") ? atoi(getenv("HRN_LONG_FACTOR")) : 4;
    double long_share = getenv("HRN_LONG_SHARE") ? atof(getenv("HRN_LONG_SHARE")) : 0.85;
    int head_slabs = 1, head_slab_px = 1024;
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

    // ---------------------------------------------------------------- planning
    int new_tensor(int ch, int h, int w) {
        Tensor t;
        t.c = ch, t.h = h, t.w = w, t.wp = w + 1, t.hp = h + 1, t.hpwp = t.wp * t.hp;
        for (size_t i = 0; i < buffers.size(); ++i)
            if (!buffers[i].in_use && buffers[i].c == ch && buffers[i].h == h && buffers[i].w == w) {
                t.buf = (int)i;
                break;
            }
        if (t.buf < 0) {
            Buffer b;
            b.c = ch, b.h = h, b.w = w;
            b.lead_rows = (size_t)t.wp + 1;
            // image rows + bottom halo + one conv block of overrun (masked lanes still form addresses)
            b.rows = b.lead_rows + (size_t)max_batch * t.hpwp + t.wp + 1 + kConvBlockRows;
            b.bytes = (size_t)align_up((int64_t)(b.rows * ch * esize), 256);
            buffers.push_back(b);
            t.buf = (int)buffers.size() - 1;
        }
        buffers[t.buf].in_use = true;
        tensors.push_back(t);
        return (int)tensors.size() - 1;
    }
    void release(int t) { buffers[tensors[t].buf].in_use = false; }

    int default_nr(int cout, int stride) const {
        int nr = (cout % 64 == 0) ? 4 : (cout % 48 == 0) ? 3 : 2;
        if (dtype == HRN_BF16 && stride == 2 && cout % 96 == 0 && direct_nr6) nr = 6;  // halves the A gathers per MFMA
        return nr;
    }
    // widest cout tile every member of a set of sibling convolutions can use
    int common_nr(const std::vector<int> &couts, int stride) const {
        bool all64 = true, all48 = true, all96 = true;
        for (int co : couts) all64 &= co % 64 == 0, all48 &= co % 48 == 0, all96 &= co % 96 == 0;
        if (dtype == HRN_BF16 && stride == 2 && all96 && direct_nr6) return 6;
        return all64 ? 4 : all48 ? 3 : 2;
    }

    // output rows per tile of the stride-2 slab kernel: as many as one slab buffer holds ((2R + 1) virtual input rows of
    // 2 * wop slots of 96 bytes)
    static int s2_rows(int wop, int ho, int cin) {
        const int vrows = s2_slot_capacity(cin) / (2 * wop);
        int r = (vrows - 1) / 2;
        return r > ho ? ho : r;
    }

    int add_conv(const std::string &conv, const std::string &bn, int in_t, int cout, int k, int stride, int relu,
                 int res_t = -1, bool emit = true, int nr_override = 0, int up = 0, int up_out_t = -1) {
        const Tensor ti = tensors[in_t];  // by value: new_tensor() below may reallocate `tensors`
        ConvOp op;
        op.conv = conv, op.bn = bn, op.in_t = in_t, op.res_t = res_t;
        op.cin = ti.c, op.cout = cout, op.k = k, op.stride = stride, op.relu = relu;
        const int oh = ti.h / stride, ow = ti.w / stride;
        op.up = up;
        op.out_t = up ? up_out_t : new_tensor(cout, oh, ow);
        const int kc = dtype == HRN_BF16 ? 32 : 16;
        const int K = k * k * op.cin;
        op.kchunks = (K + kc - 1) / kc;
        op.kpad = op.kchunks * kc;
        op.nr = nr_override ? nr_override : default_nr(cout, stride);
        op.flops = 2.0 * cout * (double)K * oh * ow;
        // pipelined LDS kernel (conv3x3_lds.hip): KS = 48 / 48-cout tiles for the HRNet-W48 branch widths, KS = 32 with
        // 64-, 48- or 32-cout tiles for everything else whose channel counts are multiples of 32
        int lds_ks = 0, lds_nrb = 0;
        if (dtype == HRN_BF16 && k == 3 && stride == 1 && !disable_lds && !up) {
            // widths that are multiples of 96 (the 96 / 192 / 384-channel branches of W48): 96 couts per block, 32-channel slices.
            // At EVERY batch size (its K order differs from the other forms'); its address arithmetic is 32-bit: tensors < 4 GB.
            const int64_t out_bytes = ((int64_t)max_batch * (oh + 1) * (ow + 1) + 2 * (ow + 2) + 512) * std::max(op.cin, cout) * 2;
            if (!disable_n96 && op.cin % 32 == 0 && cout % 96 == 0 && conv3x3_lds_bm(32, 6, ow + 1) > 0 && out_bytes < (int64_t(1) << 32))
                lds_ks = 32, lds_nrb = 6, op.n96 = true;
            else if (op.cin % 48 == 0 && cout % 48 == 0)
                lds_ks = 48, lds_nrb = 3;
            else if (op.cin % 32 == 0 && !disable_lds32)
                lds_ks = 32, lds_nrb = cout % 64 == 0 ? 4 : cout % 48 == 0 ? 3 : cout % 32 == 0 ? 2 : 0;
            if (lds_nrb && conv3x3_lds_bm(lds_ks, lds_nrb, ow + 1) == 0) lds_nrb = 0;
        }
        if (lds_nrb) {
            op.algo = 1, op.ks = lds_ks, op.nr = lds_nrb;
            op.slices = op.cin / lds_ks, op.ntiles = cout / (16 * lds_nrb), op.nch = (9 * lds_ks + 31) / 32;
            op.kpad = op.nch * 32 * op.slices;
        }
        // fp32 (the parity mode): the LDS-staged fp32 kernel (conv3x3_f32.hip) -- 16-channel slices, one K chunk of 16 per tap,
        // 48- or 32-cout tiles; everything whose channel counts allow it, at every batch size (its K order is its own)
        if (dtype == HRN_F32 && k == 3 && stride == 1 && !up && !disable_lds && !disable_f32lds && op.cin % 16 == 0 &&
            (cout % 48 == 0 || cout % 32 == 0) && conv3x3_lds_bm(16, cout % 48 == 0 ? 3 : 2, ow + 1) > 0) {
            op.algo = 1, op.ks = 16, op.nr = cout % 48 == 0 ? 3 : 2;
            op.slices = op.cin / 16, op.ntiles = cout / (16 * op.nr), op.nch = 9;
            op.kpad = 9 * 16 * op.slices;
        }
        if (dtype == HRN_BF16 && k == 3 && stride == 2 && (op.cin == 48 || op.cin == 32 || op.cin == 64) &&   // (cin = 96 does not fit the registers: conv_s2.hip)
            cout % (16 * s2_frags_per_part(op.cin)) == 0 && !up && !disable_s2 && op.algo == 0 &&
            // at least two output rows per tile (or the whole image): with one, half of every slab is halo (three input rows
            // for one output row -- the 64 -> 64 stem conv of a 384x288 net) and the kernel moves 1.5x the tensor
            (s2_rows(ow + 1, oh, op.cin) >= 2 || s2_rows(ow + 1, oh, op.cin) == oh))
            op.s2 = true;
        convs.push_back(op);
        if (emit) emit_convs({(int)convs.size() - 1});
        return op.out_t;
    }

    // the (48, 3) and the (32, 6) form live in one kernel (conv3x3_lds_kernel<48, 3>) and share launches
    static int c3_family(const ConvOp &cv) { return ((cv.ks == 48 && cv.nr == 3) || cv.n96) ? 0 : cv.ks * 16 + cv.nr; }

    // emit a set of mutually independent convolutions: one grouped launch when all of them run on the
    // LDS-staged kernel, individual launches otherwise
    // convolutions on the generic kernel -> launches appended to `dst`: one grouped launch per cout-tile width
    void emit_direct(const std::vector<int> &idx, std::vector<Op> &dst) {
        std::vector<std::vector<int>> dsets;
        for (int i : idx) {
            if (disable_dgroup) {
                dst.push_back({OP_CONV, i});
                continue;
            }
            bool placed = false;
            for (auto &set : dsets)
                if (convs[set[0]].nr == convs[i].nr && set.size() < 255) {
                    set.push_back(i);
                    placed = true;
                    break;
                }
            if (!placed) dsets.push_back({i});
        }
        for (auto &set : dsets) {
            if (set.size() == 1) {
                dst.push_back({OP_CONV, set[0]});
                continue;
            }
            DirectGroup g;
            g.conv_idx = set, g.nr = convs[set[0]].nr;
            dgroups.push_back(g);
            dst.push_back({OP_CONV_GROUP, (int)dgroups.size() - 1});
        }
    }

    void emit_convs(const std::vector<int> &idx) {
        std::vector<int> lds, direct, s2;
        for (int i : idx) {
            if (convs[i].algo == 1)
                lds.push_back(i);
            else if (convs[i].s2)
                s2.push_back(i);
            else
                direct.push_back(i);
        }
        emit_direct(direct, ops);
        if (!s2.empty()) {
            // stride-2 slab kernel: one problem per input tensor (at most 8 parts of 48 couts each), all of them one launch
            S2Group g;
            g.conv_idx = s2;
            for (int i : s2) {
                const ConvOp &cv = convs[i];
                for (int t = 0; t < cv.cout / (16 * s2_frags_per_part(cv.cin)); ++t) {
                    S2Group::Prob *pr = nullptr;
                    for (auto &q : g.probs)
                        if (q.in_t == cv.in_t && (int)q.parts.size() < kS2MaxParts) pr = &q;
                    if (!pr) {
                        g.probs.push_back(S2Group::Prob());
                        pr = &g.probs.back();
                        pr->in_t = cv.in_t;
                        const Tensor &to = tensors[cv.out_t];
                        pr->rows = s2_rows(to.wp, to.h, cv.cin);
                        pr->tiles_per_image = (to.h + pr->rows - 1) / pr->rows;
                    }
                    pr->parts.push_back({i, t});
                }
            }
            emit_direct(s2, g.fallback);
            s2groups.push_back(g);
            ops.push_back({OP_S2_GROUP, (int)s2groups.size() - 1});
        }
        if (lds.empty()) return;
        std::vector<std::vector<int>> sets;  // one launch per (KS, NRB) configuration
        for (int i : lds) {
            bool placed = false;
            if (!disable_group)
                for (auto &set : sets)
                    if (c3_family(convs[set[0]]) == c3_family(convs[i])) {
                        set.push_back(i);
                        placed = true;
                        break;
                    }
            if (!placed) sets.push_back({i});
        }
        for (auto &set : sets) {
            Conv3Group g;
            g.conv_idx = set;
            groups.push_back(g);
            ops.push_back({OP_CONV3_GROUP, (int)groups.size() - 1});
        }
    }

    int add_fuse(const std::vector<int> &terms, const std::vector<int> &shifts, const std::string &name) {
        FuseOp f;
        f.name = name;
        f.nterms = (int)terms.size();
        for (int i = 0; i < f.nterms; ++i) f.term_t[i] = terms[i], f.shift[i] = shifts[i];
        const Tensor &t0 = tensors[terms[0]];
        // output geometry = geometry of a shift-0 term
        int ref = -1;
        for (int i = 0; i < f.nterms; ++i)
            if (shifts[i] == 0) ref = terms[i];
        const Tensor &tr = tensors[ref >= 0 ? ref : terms[0]];
        (void)t0;
        f.out_t = new_tensor(tr.c, tr.h, tr.w);
        fuses.push_back(f);
        return f.out_t;
    }

    // StageModule, hrnet.py:7-71
    void add_stage(const std::string &name, std::vector<int> &xs, int nout) {
        const int nb = (int)xs.size();
        char buf[160];
        // BasicBlock x4 per branch (modules.py:56-72); the branches are independent until the fuse, so the
        // k-th conv1 (then conv2) of every branch goes out as one grouped launch
        for (int k = 0; k < 4; ++k) {
            std::vector<int> t1(nb), g1, g2;
            for (int b = 0; b < nb; ++b) {
                snprintf(buf, sizeof buf, "%s.branches.%d.%d", name.c_str(), b, k);
                const std::string p = buf;
                t1[b] = add_conv(p + ".conv1", p + ".bn1", xs[b], c << b, 3, 1, 1, -1, false);
                g1.push_back((int)convs.size() - 1);
            }
            emit_convs(g1);
            for (int b = 0; b < nb; ++b) {
                snprintf(buf, sizeof buf, "%s.branches.%d.%d", name.c_str(), b, k);
                const std::string p = buf;
                const int t2 = add_conv(p + ".conv2", p + ".bn2", t1[b], c << b, 3, 1, 1, xs[b], false);
                const int i2 = (int)convs.size() - 1, i1 = g1[b];
                // a 48 -> 48 -> 48 block in bf16: both convolutions in one pass over the tile, Y stays in LDS
                const ConvOp &a1 = convs[i1], &a2 = convs[i2];
                if (!disable_bbf && dtype == 1 && a1.algo == 1 && a2.algo == 1 && a1.ks == 48 && a1.nr == 3 && a1.slices == 1 &&
                    a1.ntiles == 1 && a2.slices == 1 && a2.ntiles == 1 && conv3x3_lds_bbf_ok(tensors[t2].wp)) {
                    convs[i1].fuse_with = i2;
                    convs[i2].fused_away = true;
                }
                g2.push_back(i2);
                release(t1[b]);
                release(xs[b]);
                xs[b] = t2;
            }
            emit_convs(g2);
        }
        // fuse layers (hrnet.py:25-51, 60-69).  The convolutions feeding the sums are issued level by level -- level 0 =
        // every 1x1 conv and the first conv of every stride-2 chain, level k = the k-th conv of the chains -- so that
        // each level is a handful of grouped launches in which siblings reading the same branch share L2.
        std::vector<std::vector<int>> term(nout, std::vector<int>(nb, -1));
        std::vector<std::vector<int>> cur(nout, std::vector<int>(nb, -1));
        for (int level = 0; level < nb; ++level) {
            std::vector<int> made, dead;
            for (int j = 0; j < nb; ++j) {
                std::vector<int> c1, c2;  // couts of the level's 1x1 / stride-2 readers of branch j
                if (level == 0)
                    for (int i = 0; i < nout && i < j; ++i) c1.push_back(c << i);
                for (int i = j + 1 + level; i < nout; ++i)
                    if (level == 0) c2.push_back(i - j == 1 ? (c << i) : (c << j));
                const int nr1 = (dgroup_nr_mode == 1 && c1.size() > 1) ? common_nr(c1, 1) : 0;
                const int nr2 = (dgroup_nr_mode == 1 && c2.size() > 1) ? common_nr(c2, 2) : 0;
                for (int i = 0; i < nout; ++i) {
                    snprintf(buf, sizeof buf, "%s.fuse_layers.%d.%d", name.c_str(), i, j);
                    const std::string q = buf;
                    if (i < j && level == 0) {  // 1x1 conv + BN, upsample folded into the fuse read (hrnet.py:30-35)
                        term[i][j] = add_conv(q + ".0", q + ".1", xs[j], c << i, 1, 1, 0, -1, false, nr1);
                        made.push_back((int)convs.size() - 1);
                    } else if (i > j && level < i - j) {  // chain of 3x3 s2 convs (hrnet.py:36-51)
                        const bool last = (level == i - j - 1);
                        snprintf(buf, sizeof buf, "%s.%d", q.c_str(), level);
                        const std::string qq = buf;
                        const int src = level == 0 ? xs[j] : cur[i][j];
                        const int nt = add_conv(qq + ".0", qq + ".1", src, last ? (c << i) : (c << j), 3, 2, last ? 0 : 1, -1,
                                                false, level == 0 ? nr2 : 0);
                        made.push_back((int)convs.size() - 1);
                        if (level > 0) dead.push_back(src);
                        cur[i][j] = nt;
                        if (last) term[i][j] = nt;
                    }
                }
            }
            if (made.empty()) break;
            emit_convs(made);
            for (int t : dead) release(t);  // only now: members of one launch must not recycle each other's inputs
        }
        std::vector<int> outs;
        for (int i = 0; i < nout; ++i) {
            std::vector<int> terms, shifts;
            for (int j = 0; j < nb; ++j) {
                terms.push_back(i == j ? xs[j] : term[i][j]);
                shifts.push_back(i < j ? j - i : 0);
            }
            snprintf(buf, sizeof buf, "%s.fuse.%d", name.c_str(), i);
            outs.push_back(add_fuse(terms, shifts, buf));
        }
        {   // the module's outputs are independent: one launch for all of them (HRN_DISABLE_FGROUP: one each)
            const int first = (int)fuses.size() - nout;
            for (int i = 0; i < nout; ++i) {
                fuses[first + i].group = disable_fgroup ? 1 : (i == 0 ? nout : 0);
                if (fuses[first + i].group) ops.push_back({OP_FUSE, first + i});
            }
        }
        for (int i = 0; i < nout; ++i)
            for (int j = 0; j < nb; ++j)
                if (i != j) release(term[i][j]);
        for (int b = 0; b < nb; ++b) release(xs[b]);
        xs = outs;
    }

    // `nblocks` Bottlenecks of 64 planes on a 64-channel input at stride 1 ("layer1" of HRNet and of PoseResNet,
    // modules.py:20-40): in bf16 conv3 of block b and conv1 of block b+1 share the chain kernel
    int add_layer1(int x, int nblocks) {
        char buf[96];
        int o1_next = -1;  // conv1 output of the next Bottleneck when the previous one already produced it
        for (int b = 0; b < nblocks; ++b) {
            snprintf(buf, sizeof buf, "layer1.%d", b);
            const std::string p = buf;
            int o1 = o1_next, r = x;
            const bool chain = dtype == HRN_BF16 && !disable_chain && b < nblocks - 1;
            int ds_idx = -1;
            if (b == 0 && chain && !disable_chain_ds) {
                // the projection shortcut is computed inside the chain kernel: its 256-channel tensor never exists
                o1 = add_conv(p + ".conv1", p + ".bn1", x, 64, 1, 1, 1);
                r = add_conv(p + ".downsample.0", p + ".downsample.1", x, 256, 1, 1, 0, -1, false, 2);
                ds_idx = (int)convs.size() - 1;
                release(r);  // never written: hand the buffer back at once
            } else if (b == 0) {  // conv1 and the projection shortcut both read x: one launch
                o1 = add_conv(p + ".conv1", p + ".bn1", x, 64, 1, 1, 1, -1, false);
                r = add_conv(p + ".downsample.0", p + ".downsample.1", x, 256, 1, 1, 0, -1, false);
                emit_convs({(int)convs.size() - 2, (int)convs.size() - 1});
            } else if (o1 < 0) {
                o1 = add_conv(p + ".conv1", p + ".bn1", x, 64, 1, 1, 1);
            }
            const int o2 = add_conv(p + ".conv2", p + ".bn2", o1, 64, 3, 1, 1);
            int o3;
            o1_next = -1;
            if (chain) {
                // conv3 (+shortcut, ReLU) of this block and conv1 (+ReLU) of the next in one pass: the 256-channel
                // tensor is written once and never read back by a 1x1 conv (bottleneck_chain.hip)
                snprintf(buf, sizeof buf, "layer1.%d", b + 1);
                const std::string pn = buf;
                o3 = add_conv(p + ".conv3", p + ".bn3", o2, 256, 1, 1, 1, r, false, 2);
                const int i3 = (int)convs.size() - 1;
                o1_next = add_conv(pn + ".conv1", pn + ".bn1", o3, 64, 1, 1, 1, -1, false, 4);
                chains.push_back({i3, (int)convs.size() - 1, ds_idx});
                ops.push_back({OP_CHAIN, (int)chains.size() - 1});
            } else {
                o3 = add_conv(p + ".conv3", p + ".bn3", o2, 256, 1, 1, 1, r);
            }
            release(o1), release(o2);
            if (b == 0 && ds_idx < 0) release(r);
            release(x);
            x = o3;
        }
        return x;
    }

    // PoseResNet (models_/poseresnet.py:16-122, Bottleneck sizes): 7x7 stem, max-pool, four ResNet layers on the
    // generic / LDS-staged conv kernels, three ConvTranspose2d + BN + ReLU as four 3x3 phase convolutions each, head
    void build_plan_poseresnet() {
        static const int spec50[4] = {3, 4, 6, 3}, spec101[4] = {3, 4, 23, 3}, spec152[4] = {3, 8, 36, 3};
        const int *layers = c == 50 ? spec50 : c == 101 ? spec101 : spec152;
        stem_out_t = new_tensor(64, H / 2, W / 2);
        ops.push_back({OP_STEM7, 0});
        int x = new_tensor(64, H / 4, W / 4);
        pool_in_t = stem_out_t, pool_out_t = x;
        ops.push_back({OP_MAXPOOL, 0});
        release(stem_out_t);
        char buf[96];
        x = add_layer1(x, layers[0]);
        for (int li = 1; li < 4; ++li) {
            const int planes = 64 << li;
            for (int b = 0; b < layers[li]; ++b) {
                snprintf(buf, sizeof buf, "layer%d.%d", li + 1, b);
                const std::string p = buf;
                const int stride = (b == 0 && li > 0) ? 2 : 1;
                int r = x;
                const int o1 = add_conv(p + ".conv1", p + ".bn1", x, planes, 1, 1, 1, -1, false);
                std::vector<int> first{(int)convs.size() - 1};
                if (b == 0) {  // projection shortcut (poseresnet.py:53-59): reads x like conv1
                    r = add_conv(p + ".downsample.0", p + ".downsample.1", x, planes * 4, 1, stride, 0, -1, false);
                    first.push_back((int)convs.size() - 1);
                }
                emit_convs(first);
                const int o2 = add_conv(p + ".conv2", p + ".bn2", o1, planes, 3, stride, 1);
                const int o3 = add_conv(p + ".conv3", p + ".bn3", o2, planes * 4, 1, 1, 1, r);
                release(o1), release(o2);
                if (b == 0) release(r);
                release(x);
                x = o3;
            }
        }
        for (int i = 0; i < 3; ++i) {  // deconv_layers: ConvTranspose2d(4, s2, p1) + BN + ReLU (poseresnet.py:84-104)
            const Tensor tx = tensors[x];
            const int up_t = new_tensor(256, tx.h * 2, tx.w * 2);
            snprintf(buf, sizeof buf, "deconv_layers.%d", 3 * i);
            const std::string cn = buf;
            snprintf(buf, sizeof buf, "deconv_layers.%d", 3 * i + 1);
            const std::string bn = buf;
            std::vector<int> phases;
            for (int ph = 0; ph < 4; ++ph) {
                add_conv(cn, bn, x, 256, 2, 1, 1, -1, false, 0, 1 + ph, up_t);  // k = 2: the four live taps of the phase
                phases.push_back((int)convs.size() - 1);
            }
            emit_convs(phases);
            release(x);
            x = up_t;
        }
        head_in_t = x;
        head_c = tensors[x].c;
        ops.push_back({OP_HEAD, 0});
        ops.push_back({OP_DECODE, 0});
        layout_blob();
    }

    void build_plan() {
        if (model == 1) return build_plan_poseresnet();
        head_c = c;
        // stem conv1 (dedicated kernel), hrnet.py:158-160
        stem_out_t = new_tensor(64, H / 2, W / 2);
        ops.push_back({OP_STEM, 0});
        int x = add_conv("conv2", "bn2", stem_out_t, 64, 3, 2, 1);  // hrnet.py:161-163
        release(stem_out_t);
        char buf[96];
        x = add_layer1(x, 4);  // layer1: Bottleneck x4, modules.py:20-40
        std::vector<int> xs;
        xs.push_back(add_conv("transition1.0.0", "transition1.0.1", x, c, 3, 1, 1, -1, false));
        xs.push_back(add_conv("transition1.1.0.0", "transition1.1.0.1", x, 2 * c, 3, 2, 1, -1, false));
        emit_convs({(int)convs.size() - 2, (int)convs.size() - 1});
        release(x);
        add_stage("stage2.0", xs, 2);
        xs.push_back(add_conv("transition2.2.0.0", "transition2.2.0.1", xs[1], 4 * c, 3, 2, 1));
        for (int m = 0; m < 4; ++m) {
            snprintf(buf, sizeof buf, "stage3.%d", m);
            add_stage(buf, xs, 3);
        }
        xs.push_back(add_conv("transition3.3.0.0", "transition3.3.0.1", xs[2], 8 * c, 3, 2, 1));
        add_stage("stage4.0", xs, 4);
        add_stage("stage4.1", xs, 4);
        add_stage("stage4.2", xs, 1);
        head_in_t = xs[0];
        ops.push_back({OP_HEAD, 0});
        ops.push_back({OP_DECODE, 0});

        layout_blob();
    }

    // Every tensor a launch of the pass writes to HBM, by name (hrn_forward_tap): "stem" (conv1 + bn1 + ReLU), every
    // convolution under its state_dict prefix (its output after bias / residual / ReLU, as stored), "<stage>.fuse.<i>".
    // A tensor stays intact at least until the op after the one that completed it: buffers are recycled by LATER tensors only.
    void index_taps() {
        taps.clear();
        auto add = [&](const std::string &name, int tensor, int op, int conv) {
            for (TapPoint &t : taps)
                if (t.name == name) {  // the four phases of a transposed convolution write one tensor: the last launch completes it
                    t.op = op;
                    return;
                }
            taps.push_back(TapPoint{name, tensor, op, conv});
        };
        for (size_t oi = 0; oi < ops.size(); ++oi) {
            const Op &op = ops[oi];
            switch (op.kind) {
                case OP_STEM:
                case OP_STEM7: add("stem", stem_out_t, (int)oi, -1); break;
                case OP_MAXPOOL: add("maxpool", pool_out_t, (int)oi, -1); break;
                case OP_CONV: add(convs[op.idx].conv, convs[op.idx].out_t, (int)oi, op.idx); break;
                case OP_CONV3_GROUP:
                    for (int ci : groups[op.idx].conv_idx) add(convs[ci].conv, convs[ci].out_t, (int)oi, ci);
                    break;
                case OP_CONV_GROUP:
                    for (int ci : dgroups[op.idx].conv_idx) add(convs[ci].conv, convs[ci].out_t, (int)oi, ci);
                    break;
                case OP_S2_GROUP:
                    for (int ci : s2groups[op.idx].conv_idx) add(convs[ci].conv, convs[ci].out_t, (int)oi, ci);
                    break;
                case OP_CHAIN:  // (the projection shortcut folded into the chain kernel is never written)
                    add(convs[chains[op.idx].conv3].conv, convs[chains[op.idx].conv3].out_t, (int)oi, chains[op.idx].conv3);
                    add(convs[chains[op.idx].conv1].conv, convs[chains[op.idx].conv1].out_t, (int)oi, chains[op.idx].conv1);
                    break;
                case OP_FUSE:
                    for (int k = 0; k < fuses[op.idx].group; ++k) add(fuses[op.idx + k].name, fuses[op.idx + k].out_t, (int)oi, -1);
                    break;
                default: break;
            }
        }
    }

    void layout_blob() {
        // weight blob layout
        int64_t off = 0;
        stem_w_off = off, off = align_up(off + (model == 1 ? 147 : 27) * 64 * 4, 256);  // PoseResNet: 7x7 stem
        stem_b_off = off, off = align_up(off + 64 * 4, 256);
        stem_wp_off = off, off = align_up(off + (model == 1 ? 5 : 1) * 4 * 1024, 256);  // bf16 MFMA image of conv1 (stem_mfma_kernel / stem7_mfma_kernel)
        for (auto &cv : convs) {
            cv.w_off = off;
            cv.w_bytes = cv.algo == 1 ? (int64_t)cv.ntiles * cv.slices * cv.nch * cv.nr * 1024
                                      : (int64_t)(cv.cout / 16) * cv.kchunks * 1024;
            off = align_up(off + cv.w_bytes, 256);
            cv.b_off = off;
            off = align_up(off + cv.cout * 4, 256);
            if (cv.s2) {  // the slab kernel's image: [cout tile][K chunks][frags][64 lanes][16 B], (48, 3): 14 chunks, (96, 2): 27
                const int nf = s2_frags_per_part(cv.cin);
                cv.w2_off = off;
                cv.w2_bytes = (int64_t)(cv.cout / (16 * nf)) * ((9 * cv.cin + 31) / 32) * nf * 1024;
                off = align_up(off + cv.w2_bytes, 256);
            }
        }
        head_w_off = off, off = align_up(off + (int64_t)joints * head_c * 4, 256);
        head_b_off = off, off = align_up(off + joints * 4, 256);
        head_wp_off = off, off = align_up(off + 2 * ((head_c + 31) / 32) * 1024, 256);  // bf16 MFMA image (head_mfma_kernel)
        blob_bytes = off;

        const int hw = (H / 4) * (W / 4);
        head_slab_px = 1024;
        head_slabs = (hw + head_slab_px - 1) / head_slab_px;
    }

    // ---------------------------------------------------------------- device memory
    bool hip_ok(hipError_t e, const char *what) {
        if (e == hipSuccess) return true;
        err = std::string(what) + ": " + hipGetErrorString(e);
        return false;
    }

    bool allocate() {
        workspace_bytes = 0;
        for (auto &b : buffers) workspace_bytes += (int64_t)b.bytes;
        const int64_t part = (int64_t)max_batch * joints * head_slabs;
        workspace_bytes += part * 8;
        if (plan_only) {
            fill_problems();
            index_s2groups();
            blob = (char *)calloc(1, (size_t)blob_bytes);
            return blob != nullptr;
        }
        if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return false;
        for (auto &b : buffers) {
            if (!hip_ok(hipMalloc((void **)&b.dev, b.bytes), "hipMalloc(activation)")) return false;
            if (!hip_ok(hipMemset(b.dev, 0, b.bytes), "hipMemset(activation)")) return false;
        }
        if (!hip_ok(hipMalloc((void **)&blob, (size_t)blob_bytes), "hipMalloc(weights)")) return false;
        if (!hip_ok(hipMemset(blob, 0, (size_t)blob_bytes), "hipMemset(weights)")) return false;
        if (!setup_groups() || !setup_dgroups() || !setup_s2groups()) return false;
        if (!hip_ok(hipMalloc((void **)&part_val, (size_t)part * 4), "hipMalloc(part_val)")) return false;
        if (!hip_ok(hipMalloc((void **)&part_idx, (size_t)part * 4), "hipMalloc(part_idx)")) return false;
        return hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize");
    }

    // every block walks ~`half_stages_per_block` half-slices so that blocks of all branches last alike
    int conv3_tiles_per_block(const ConvOp &cv) const {
        // (96-cout form: three stages of 18 MR MFMAs per slice, each about 0.86 of a (48, 3) half-stage)
        if (cv.n96) return std::max(1, (half_stages_per_block * 7) / (6 * 3 * cv.slices));
        const int t = half_stages_per_block / ((cv.ks == 48 ? 2 : 1) * cv.slices);
        return t < 1 ? 1 : t;
    }

    // A fusable BasicBlock runs fused (conv1's launch does both convolutions, conv2's launch skips it) when the call is
    // large enough; both launches decide by the same rule.
    bool bbf_active(const ConvOp &cv, int nb) const {
        return (nb * tensors[cv.out_t].hpwp + 511) / 512 >= bbf_min_tiles;
    }
    bool fused_now(const ConvOp &cv, int nb) const { return cv.fuse_with >= 0 && bbf_active(cv, nb); }
    bool skipped(const ConvOp &cv, int nb) const { return cv.fused_away && bbf_active(cv, nb); }

    // device-resident descriptors + block maps of the grouped conv launches
    int group_blocks(const Conv3Group &g, int nb, std::vector<int2> *out, bool reverse, bool all_short = false,
                     std::vector<int> *tile_px_out = nullptr) const {
        struct Ent {
            double key;
            int2 v;
            int bm;
        };
        std::vector<Ent> ents;
        // Block lengths follow the size of the launch: small batches get shorter blocks (at least ~2 per CU before
        // anything else), and long blocks are only worth it when there are many blocks per CU to begin with.
        // a few crops: even one tile per block leaves CUs idle -> 128-pixel tiles for the whole launch
        long one_per_block = 0;
        for (int ci : g.conv_idx) {
            const ConvOp &cv = convs[ci];
            if (skipped(cv, nb)) continue;
            const Tensor &to = tensors[cv.out_t];
            const int bmn = fused_now(cv, nb) ? 512 : conv3x3_lds_bm(cv.ks, cv.nr, to.wp);
            one_per_block += (long)((nb * to.hpwp + bmn - 1) / bmn) * cv.ntiles;
        }
        const bool small = small_tiles && one_per_block < small_below;
        // a fused BasicBlock walks 512-pixel tiles through both convolutions (no small-tile mode): about three
        // ordinary 384-pixel tiles' worth of work each
        // fp32 form: a stage (one 16-channel slice of a 512-pixel tile) keeps a CU busy for ~8 us, so a one-tile block of the
        // 256-channel branch would run 16 of them back to back while the launch as a whole is ~20 stages of work per CU:
        // deep convolutions take 128-pixel tiles (the kernel picks the tile size per block) so that no block exceeds ~16
        // quarter-stages and the launch can balance
        auto small_conv = [&](const ConvOp &cv) { return small || (cv.ks == 16 && f32_small_slices > 0 && cv.slices >= f32_small_slices); };
        auto tile_px = [&](const ConvOp &cv, const Tensor &to) {
            return fused_now(cv, nb) ? 512 : small_conv(cv) ? 128 : conv3x3_lds_bm(cv.ks, cv.nr, to.wp);
        };
        auto tiles_per_block = [&](const ConvOp &cv, int div) {
            return std::max(1, conv3_tiles_per_block(cv) / (fused_now(cv, nb) ? bbf_tpb_div : 1) / div);
        };
        auto count_blocks = [&](int div) {
            long total = 0;
            for (int ci : g.conv_idx) {
                const ConvOp &cv = convs[ci];
                if (skipped(cv, nb)) continue;
                const Tensor &to = tensors[cv.out_t];
                const int bm = tile_px(cv, to);
                const int mtiles = (nb * to.hpwp + bm - 1) / bm;
                const int tpb = tiles_per_block(cv, div);
                total += (long)((mtiles + tpb - 1) / tpb) * cv.ntiles;
            }
            return total;
        };
        int div = 1;
        while (div < 64 && count_blocks(div) < 512) div *= 2;
        int lf = all_short ? 1 : (int)std::min<long>(long_factor, count_blocks(div) / 768);
        if (lf < 1) lf = 1;
        for (size_t k = 0; k < g.conv_idx.size(); ++k) {
            const ConvOp &cv = convs[g.conv_idx[k]];
            if (skipped(cv, nb)) continue;
            const Tensor &to = tensors[cv.out_t];
            const int bm = tile_px(cv, to);
            const bool fused = fused_now(cv, nb);
            const int prob = fused ? g.fused_prob[k] : (int)k;
            const int mtiles = (nb * to.hpwp + bm - 1) / bm;
            // Blocks come in two lengths: long ones (fewer pipeline prologues -- a block's first loads have nothing to
            // hide behind) over the first `long_share` of the M tiles, short ones over the rest to fill the tail.
            const int tpb_short = tiles_per_block(cv, div);
            const int tpb_long = tpb_short * lf;
            const int long_tiles = ((int)(mtiles * long_share) / (8 * tpb_long)) * (8 * tpb_long);  // whole XCD rounds
            for (int phase = 0; phase < 2; ++phase) {
                const int tpb = phase == 0 ? tpb_long : tpb_short;
                const int first = phase == 0 ? 0 : long_tiles;
                const int count = phase == 0 ? long_tiles : mtiles - long_tiles;
                if (count <= 0) continue;
                const int mgroups = (count + tpb - 1) / tpb;
                const int total = mgroups * cv.ntiles;
                // XCD-aware order inside a problem: the hardware places block id b on XCD b % 8 (private L2 each).
                // Emit rounds of 8 M groups x all cout tiles with the M group varying fastest, so the cout tiles of one
                // M group are 8 ids apart = on the same XCD, and re-read its slab from that L2, not over the fabric.
                int seq = 0;
                for (int round = 0; round * 8 < mgroups; ++round)
                    for (int nt = 0; nt < cv.ntiles; ++nt)
                        for (int x = 0; x < 8; ++x) {
                            const int mg = round * 8 + x;
                            if (mg >= mgroups) continue;
                            const int i = seq++;
                            int tiles = count - mg * tpb;
                            if (tiles > tpb) tiles = tpb;
                            double key = (i + 0.5) / total;  // proportional interleave of the problems
                            if (block_order == 1)            // longest-processing-time first (estimated block cost)
                                key = -(double)tiles * (fused         ? 28000.0
                                                        : cv.ks == 16 ? cv.slices * (double)bm * 36.0 + 2000.0   // fp32: 9 x 4 MR NRB MFMAs of 32 cycles per slice
                                                        : cv.n96      ? cv.slices * 3.0 * (bm == 512 ? 3900.0 : 3000.0) + (bm == 512 ? 7000.0 : 5000.0)
                                                                 : cv.slices * 2.0 * (bm == 512 ? 4300.0 : 3500.0) + (bm == 512 ? 5000.0 : 3000.0)) +
                                      1e-3 * key;
                            int mt0 = first + mg * tpb;
                            // every other launch walks the tensors backwards: a launch starts on what its producer
                            // wrote last, i.e. on the part most likely still in the Infinity Cache
                            if (reverse) mt0 = mtiles - mt0 - tiles;
                            ents.push_back({key, int2{prob | (nt << 8) | (tiles << 16), mt0 | (fused ? 1 << 29 : small_conv(cv) ? 1 << 30 : 0)}, bm});
                        }
            }
        }
        std::stable_sort(ents.begin(), ents.end(), [](const Ent &a, const Ent &b) { return a.key < b.key; });
        if (tile_px_out) {
            tile_px_out->resize(ents.size());
            for (size_t i = 0; i < ents.size(); ++i) (*tile_px_out)[i] = ents[i].bm;
        }
        if (out) {
            out->resize(ents.size());
            for (size_t i = 0; i < ents.size(); ++i) (*out)[i] = ents[i].v;
        }
        return (int)ents.size();
    }

    // The persistent work-queue form of a grouped launch for a call of nb crops (conv3x3_queue.inc): the units are the block
    // map's entries (same order: longest first, XCD-aware inside a convolution) with `queue_tpb` M tiles each; the fused
    // BasicBlock of the launch (at most one: the 48-channel branch) is not queued -- `bbf_blocks` of the CUs each take an equal
    // range of its tiles first.  Returns false when the launch has to take the per-block form: a member that is neither a
    // 96-cout-form convolution on 512-pixel tiles nor the fused block, 128-pixel tiles, too few units to pay.
    bool queue_plan(const Conv3Group &g, int nb, bool reverse, std::vector<QUnit> *units, int *bbf_prob, int *bbf_blocks, int *bbf_tiles,
                    std::vector<int> *unit_conv = nullptr, int *bbf_conv = nullptr) const {
        if (!queue_on || dtype != HRN_BF16 || probs_host.empty()) return false;
        *bbf_prob = 0, *bbf_blocks = 0, *bbf_tiles = 0;
        struct Ent {
            double key;
            QUnit u;
            int conv;
        };
        std::vector<Ent> ents;
        double cost_bbf = 0, cost_q = 0;
        if (bbf_conv) *bbf_conv = -1;
        int nfused = 0;
        for (size_t k = 0; k < g.conv_idx.size(); ++k) {
            const ConvOp &cv = convs[g.conv_idx[k]];
            if (skipped(cv, nb)) continue;
            const Tensor &to = tensors[cv.out_t];
            const int mtiles = (nb * to.hpwp + 511) / 512;
            if (fused_now(cv, nb)) {
                if (++nfused > 1) return false;
                *bbf_prob = g.fused_prob[k], *bbf_tiles = mtiles;
                if (bbf_conv) *bbf_conv = g.conv_idx[k];
                cost_bbf = mtiles * 28000.0;
                continue;
            }
            if (!cv.n96 || cv.slices < 3 || conv3x3_lds_bm(cv.ks, cv.nr, to.wp) != 512) return false;
            const Conv3Problem &pr = probs_host[g.prob_first + k];
            const double tile_cost = cv.slices * 3.0 * 3900.0 + 2000.0;
            const int tpb = queue_tpb;
            const int mgroups = (mtiles + tpb - 1) / tpb;
            int seq = 0;
            // XCD-aware order inside a convolution (see group_blocks): rounds of 8 M groups x all cout tiles, M group fastest
            for (int round = 0; round * 8 < mgroups; ++round)
                for (int nt = 0; nt < cv.ntiles; ++nt)
                    for (int x = 0; x < 8; ++x) {
                        const int mg = round * 8 + x;
                        if (mg >= mgroups) continue;
                        int tiles = std::min(tpb, mtiles - mg * tpb);
                        int mt0 = mg * tpb;
                        if (reverse) mt0 = mtiles - mt0 - tiles;
                        ents.push_back({-tiles * tile_cost + 1e-9 * seq++, make_qunit(pr, nt, mt0, tiles, nb), g.conv_idx[k]});
                        cost_q += tiles * tile_cost;
                    }
        }
        if ((long)ents.size() < (long)queue_min_units_per_cu * num_cus) return false;
        std::stable_sort(ents.begin(), ents.end(), [](const Ent &a, const Ent &b) { return a.key < b.key; });
        if (nfused) {
            int nbb = (int)(num_cus * queue_bbf_scale * cost_bbf / (cost_bbf + cost_q) / 8.0 + 0.5) * 8;   // whole XCD rounds
            *bbf_blocks = std::max(8, std::min(num_cus - 8, nbb));
        }
        if (units) {
            units->resize(ents.size());
            for (size_t i = 0; i < ents.size(); ++i) (*units)[i] = ents[i].u;
        }
        if (unit_conv) {
            unit_conv->resize(ents.size());
            for (size_t i = 0; i < ents.size(); ++i) (*unit_conv)[i] = ents[i].conv;
        }
        return true;
    }

    // the pixel grid a convolution iterates over: its output, or for a transposed-conv phase its input
    const Tensor &grid_of(const ConvOp &cv) const { return tensors[cv.up ? cv.in_t : cv.out_t]; }

    ConvArgs conv_args(const ConvOp &cv, int nb, bool rev) const {
        const Tensor &ti = tensors[cv.in_t], &to = grid_of(cv);
        ConvArgs a;
        a.in = row0(cv.in_t), a.out = row0(cv.out_t);
        a.w = blob + cv.w_off, a.bias = (const float *)(blob + cv.b_off);
        a.res = cv.res_t >= 0 ? row0(cv.res_t) : nullptr;
        a.cin = cv.cin, a.cout = cv.cout;
        a.in_wp = ti.wp, a.in_hpwp = ti.hpwp;
        a.out_h = to.h, a.out_w = to.w, a.out_wp = to.wp, a.out_hpwp = to.hpwp;
        a.m = nb * to.hpwp;
        a.ksize = cv.k, a.stride = cv.stride, a.relu = cv.relu, a.kchunks = cv.kchunks;
        a.rev = rev;
        a.wlds = direct_wlds && dtype == 1 ? 1 : 0;
        a.up = cv.up ? 1 : 0, a.up_a = cv.up ? (cv.up - 1) >> 1 : 0, a.up_b = cv.up ? (cv.up - 1) & 1 : 0;
        a.up_wp = tensors[cv.out_t].wp, a.up_hpwp = tensors[cv.out_t].hpwp;
        for (int t = 0; t < 4; ++t) {  // live tap t = ty*2 + tx of phase (a, b): dy = ty - 1 + a, dx = tx - 1 + b
            const int dy = (t >> 1) - 1 + a.up_a, dx = (t & 1) - 1 + a.up_b;
            a.taps[t] = cv.up ? dy * ti.wp + dx : 0;
        }
        return a;
    }

    // Block map of a grouped launch of the generic kernel.  Convolutions that read the same tensor with the same
    // window form a class; inside a class the map walks rounds of 8 M tiles and, per round, every (conv, cout tile)
    // of the class with the M tile varying fastest: block id b runs on XCD b % 8, so all readers of one input tile
    // are 8 ids apart = on one XCD, back to back, and the tile comes from HBM once.  Rounds are padded to 8 entries
    // (an out-of-range M tile returns at once) to keep that alignment.  Classes go out costliest first.
    int direct_group_blocks(const DirectGroup &g, int nb, std::vector<int2> *out, int mr = 4) const {
        struct Cls {
            int in_t, k, stride;
            std::vector<int> members;
            double cost = 0;
        };
        std::vector<Cls> cls;
        for (size_t k = 0; k < g.conv_idx.size(); ++k) {
            const ConvOp &cv = convs[g.conv_idx[k]];
            size_t ci = 0;
            for (; ci < cls.size(); ++ci)
                if (cls[ci].in_t == cv.in_t && cls[ci].k == cv.k && cls[ci].stride == cv.stride) break;
            if (ci == cls.size()) cls.push_back(Cls{cv.in_t, cv.k, cv.stride, {}, 0});
            cls[ci].members.push_back((int)k);
            cls[ci].cost += (double)grid_of(cv).hpwp * (cv.cout / (16 * g.nr)) * cv.kchunks;
        }
        std::stable_sort(cls.begin(), cls.end(), [](const Cls &a, const Cls &b) { return a.cost > b.cost; });
        int n = 0;
        if (out) out->clear();
        for (const Cls &cl : cls) {
            const int mtiles = (nb * grid_of(convs[g.conv_idx[cl.members[0]]]).hpwp + 64 * mr - 1) / (64 * mr);
            for (int round = 0; round * 8 < mtiles; ++round)
                for (int k : cl.members) {
                    const int ngroups = convs[g.conv_idx[k]].cout / (16 * g.nr);
                    for (int ng = 0; ng < ngroups; ++ng)
                        for (int x = 0; x < 8; ++x, ++n)
                            if (out) out->push_back(int2{k | (ng << 8), round * 8 + x});
                }
        }
        return n;
    }

    bool setup_dgroups() {
        for (auto &g : dgroups) {
            // M tiles shrink (mr 4 -> 2 -> 1) only while a launch has fewer than 512 blocks: 4x that bounds every case
            g.map_capacity = std::max<int64_t>(direct_group_blocks(g, max_batch, nullptr), 4 * 512 + 64 * (int64_t)g.conv_idx.size() * 8);
            for (MapSlot &sl : g.slot) {
                if (!alloc_slot(sl, g.map_capacity, g.conv_idx.size())) return false;
                workspace_bytes += g.map_capacity * (int64_t)sizeof(int2) + (int64_t)(g.conv_idx.size() * sizeof(ConvArgs));
            }
        }
        return true;
    }

    // ---- stride-2 slab kernel: descriptors (fixed at create) and block maps (per micro-batch size)
    size_t index_s2groups() {
        size_t n = 0;
        for (auto &g : s2groups) g.prob_first = (int)n, n += g.probs.size();
        return n;
    }
    long s2_tiles(const S2Group &g, int nb) const {
        long t = 0;
        for (auto &pr : g.probs) t += (long)nb * pr.tiles_per_image;
        return t;
    }
    // the slab kernel runs when the launch has a tile for every CU; smaller calls take the generic kernel (bit-identical)
    bool s2_active(const S2Group &g, int nb) const { return s2_tiles(g, nb) >= s2_min_tiles; }
    // Block map: every block walks a run of consecutive tiles of one problem (weights loaded once per block); run lengths are
    // chosen so that blocks of all problems cost about the same and the launch has ~s2_target_blocks of them, costliest first.
    int s2_blocks(const S2Group &g, int nb, std::vector<int2> *out) const {
        struct Ent {
            double key;
            int2 v;
        };
        std::vector<double> cost(g.probs.size());
        double total = 0;
        for (size_t k = 0; k < g.probs.size(); ++k) {
            const S2Group::Prob &pr = g.probs[k];
            const Tensor &to = tensors[convs[pr.parts[0].first].out_t];
            const int frags = (pr.rows * to.wp + 15) / 16;
            const int wm = std::max(1, 8 / (int)pr.parts.size());     // fewest waves sharing a part
            const int cin = convs[pr.parts[0].first].cin;
            cost[k] = (double)((frags + wm - 1) / wm) * ((9 * cin + 31) / 32) * s2_frags_per_part(cin) + 60;   // MFMAs of the busiest wave + per-tile overhead
            total += cost[k] * nb * pr.tiles_per_image;
        }
        const double per_block = total / s2_target_blocks;
        std::vector<Ent> ents;
        for (size_t k = 0; k < g.probs.size(); ++k) {
            const S2Group::Prob &pr = g.probs[k];
            const int tiles = nb * pr.tiles_per_image;
            int run = (int)(per_block / cost[k] + 0.5);
            run = run < 1 ? 1 : run > 64 ? 64 : run;
            for (int t0 = 0; t0 < tiles; t0 += run) {
                const int cnt = tiles - t0 < run ? tiles - t0 : run;
                ents.push_back({-(cnt * cost[k]) + 1e-9 * t0, int2{(int)k | (cnt << 8), t0}});
            }
        }
        std::stable_sort(ents.begin(), ents.end(), [](const Ent &a, const Ent &b) { return a.key < b.key; });
        if (out) {
            out->resize(ents.size());
            for (size_t i = 0; i < ents.size(); ++i) (*out)[i] = ents[i].v;
        }
        return (int)ents.size();
    }

    // Eight waves over P parts: 8 / P waves per part (the first 8 % P parts one more), the waves of a part deal its pixel
    // fragments round-robin.  Wave w runs on SIMD w % 4: waves are placed so that the four SIMDs carry equal shares
    // (3 parts: 3 + 3 + 2 waves -> per-SIMD load 10 : 10 : 8 : 8 twelfths instead of 12 : 12 : 6 : 6 with 2 + 2 + 2).
    static void s2_wave_table(int nparts, unsigned char *part, unsigned char *f0, unsigned char *fs) {
        struct W {
            int p, i, m;
        };
        std::vector<W> ws;
        for (int p = 0; p < nparts; ++p) {
            const int m = std::max(1, 8 / nparts + (p < 8 % nparts ? 1 : 0));
            for (int i = 0; i < m && (int)ws.size() < 8; ++i) ws.push_back({p, i, m});
        }
        std::stable_sort(ws.begin(), ws.end(), [](const W &a, const W &b) { return a.m < b.m; });  // heaviest (fewest sharers) first
        double load[4] = {0, 0, 0, 0};
        bool used[8] = {};
        for (int w = 0; w < 8; ++w) part[w] = 0xff, f0[w] = 0, fs[w] = 1;
        for (const W &x : ws) {
            int best = -1;
            for (int w = 0; w < 8; ++w)
                if (!used[w] && (best < 0 || load[w & 3] < load[best & 3] - 1e-12)) best = w;
            used[best] = true;
            load[best & 3] += 1.0 / x.m;
            part[best] = (unsigned char)x.p, f0[best] = (unsigned char)x.i, fs[best] = (unsigned char)x.m;
        }
    }

    bool setup_s2groups() {
        const size_t nprob = index_s2groups();
        if (!nprob) return true;
        std::vector<S2Problem> hp(nprob);
        for (auto &g : s2groups) {
            for (size_t k = 0; k < g.probs.size(); ++k) {
                const S2Group::Prob &pr = g.probs[k];
                const Tensor &ti = tensors[pr.in_t], &to = tensors[convs[pr.parts[0].first].out_t];
                S2Problem &q = hp[g.prob_first + k];
                memset(&q, 0, sizeof q);
                q.in = row0(pr.in_t), q.cin = ti.c, q.in_wp = ti.wp, q.in_hpwp = ti.hpwp;
                q.ho = to.h, q.wo = to.w, q.wop = to.wp, q.out_hpwp = to.hpwp;
                q.rows = pr.rows, q.tiles_per_image = pr.tiles_per_image;
                q.nparts = (int)pr.parts.size();
                s2_wave_table(q.nparts, q.wave_part, q.wave_f0, q.wave_fs);
                fast_div(to.wp, &q.magic_wop, &q.shift_wop);
                for (int i = 0; i < q.nparts; ++i) {
                    const ConvOp &cv = convs[pr.parts[i].first];
                    S2Part &pt = q.part[i];
                    const int nf = s2_frags_per_part(cv.cin);
                    pt.w = blob + cv.w2_off + (int64_t)pr.parts[i].second * ((9 * cv.cin + 31) / 32) * nf * 1024;
                    pt.bias = (const float *)(blob + cv.b_off);
                    pt.out = row0(cv.out_t);
                    pt.cout = cv.cout, pt.ch0 = pr.parts[i].second * 16 * nf, pt.relu = cv.relu;
                }
            }
            g.map_capacity = 64;
            for (auto &pr : g.probs) g.map_capacity += (int64_t)max_batch * pr.tiles_per_image;  // one tile per block bounds every split
            for (MapSlot &sl : g.slot) {
                if (!alloc_slot(sl, g.map_capacity, 0)) return false;
                workspace_bytes += g.map_capacity * (int64_t)sizeof(int2);
            }
        }
        if (!hip_ok(hipMalloc((void **)&s2probs_dev, nprob * sizeof(S2Problem)), "hipMalloc(s2 problems)")) return false;
        return hip_ok(hipMemcpy(s2probs_dev, hp.data(), nprob * sizeof(S2Problem), hipMemcpyHostToDevice), "hipMemcpy(s2 problems)");
    }

    bool alloc_slot(MapSlot &sl, int64_t capacity, size_t nargs) {
        if (!hip_ok(hipMalloc((void **)&sl.dev, (size_t)capacity * sizeof(int2)), "hipMalloc(blockmap)")) return false;
        if (!hip_ok(hipHostMalloc((void **)&sl.pin, (size_t)capacity * sizeof(int2), hipHostMallocDefault), "hipHostMalloc(blockmap)"))
            return false;
        if (nargs) {
            if (!hip_ok(hipMalloc((void **)&sl.args_dev, nargs * sizeof(ConvArgs)), "hipMalloc(conv args)")) return false;
            if (!hip_ok(hipHostMalloc((void **)&sl.args_pin, nargs * sizeof(ConvArgs), hipHostMallocDefault), "hipHostMalloc(conv args)"))
                return false;
        }
        return hip_ok(hipEventCreateWithFlags(&sl.landed, hipEventDisableTiming), "hipEventCreate");
    }
    void free_slot(MapSlot &sl) {
        if (sl.dev) (void)hipFree(sl.dev);
        if (sl.pin) (void)hipHostFree(sl.pin);
        if (sl.args_dev) (void)hipFree(sl.args_dev);
        if (sl.args_pin) (void)hipHostFree(sl.args_pin);
        if (sl.q_dev) (void)hipFree(sl.q_dev);
        if (sl.q_pin) (void)hipHostFree(sl.q_pin);
        if (sl.landed) (void)hipEventDestroy(sl.landed);
        sl = MapSlot();
    }
    // the slot holding the maps of micro-batch size nb, or the least recently used one to rebuild (*hit = false); a
    // slot about to be rebuilt has had its previous upload waited for, so its pinned image may be rewritten
    MapSlot *find_slot(MapSlot (&slots)[kMapSlots], int nb, bool *hit) {
        MapSlot *lru = &slots[0];
        for (MapSlot &sl : slots) {
            if (sl.nb == nb) {
                sl.stamp = ++map_clock;
                *hit = true;
                return &sl;
            }
            if (sl.stamp < lru->stamp) lru = &sl;
        }
        *hit = false;
        if (lru->nb >= 0) (void)hipEventSynchronize(lru->landed);
        lru->nb = -1;
        lru->stamp = ++map_clock;
        return lru;
    }

    // a cached map was uploaded on sl->up_stream; a launch on another stream has to see that upload (callers that
    // alternate streams on one handle -- predict_stream, user code under torch.cuda.stream)
    hipError_t slot_ready(MapSlot *sl, bool hit, hipStream_t s) {
        if (hit && sl->up_stream != s) return hipStreamWaitEvent(s, sl->landed, 0);
        if (!hit) sl->up_stream = s;
        return hipSuccess;
    }

    // descriptor numbering of the grouped launches (host only; plan-only handles need it for hrn_plan_block_map)
    size_t index_groups() {
        size_t nprob = 0;
        for (auto &g : groups) {
            g.prob_first = (int)nprob;
            nprob += g.conv_idx.size();
            g.fused_prob.assign(g.conv_idx.size(), -1);
            int extra = (int)g.conv_idx.size();
            for (size_t k = 0; k < g.conv_idx.size(); ++k)
                if (convs[g.conv_idx[k]].fuse_with >= 0) g.fused_prob[k] = extra++, ++nprob;  // a second descriptor: the fused form
        }
        return nprob;
    }

    // the descriptors of the grouped launches on the host (plan-only handles: the same, with the pointers of an unallocated
    // workspace -- hrn_plan_queue reads the geometry only)
    size_t fill_problems() {
        const size_t nprob = index_groups();
        probs_host.assign(nprob, Conv3Problem{});
        std::vector<Conv3Problem> &hp = probs_host;
        for (auto &g : groups) {
            g.max_wp = 0;
            for (size_t k = 0; k < g.conv_idx.size(); ++k) {
                const ConvOp &cv = convs[g.conv_idx[k]];
                const Tensor &to = tensors[cv.out_t];
                Conv3Problem &q = hp[g.prob_first + k];
                q.in = row0(cv.in_t), q.out = row0(cv.out_t);
                q.w = blob + cv.w_off, q.bias = (const float *)(blob + cv.b_off);
                q.res = cv.res_t >= 0 ? row0(cv.res_t) : nullptr;
                q.cin = cv.cin, q.cout = cv.cout, q.h = to.h, q.wd = to.w, q.wp = to.wp, q.hpwp = to.hpwp;
                q.relu = cv.relu, q.slices = cv.slices, q.ntiles = cv.ntiles;
                q.w2 = nullptr, q.bias2 = nullptr;
                q.n96 = cv.n96 ? 1 : 0;
                q.tiles_per_block = conv3_tiles_per_block(cv);
                q.bm = conv3x3_lds_bm(cv.ks, cv.nr, to.wp);
                fast_div(to.hpwp, &q.magic_hpwp, &q.shift_hpwp);
                fast_div(to.wp, &q.magic_wp, &q.shift_wp);
                if (to.wp > g.max_wp) g.max_wp = to.wp;
                if (cv.fuse_with >= 0) {  // the whole BasicBlock: in = x (also the residual), out = the block's output
                    const ConvOp &c2 = convs[cv.fuse_with];
                    Conv3Problem &f = hp[g.prob_first + g.fused_prob[k]];
                    f = q;
                    f.out = row0(c2.out_t), f.relu = c2.relu, f.res = nullptr;
                    f.w2 = blob + c2.w_off, f.bias2 = (const float *)(blob + c2.b_off);
                }
            }
        }
        return nprob;
    }

    bool setup_groups() {
        const size_t nprob = fill_problems();
        if (!nprob) return true;
        std::vector<Conv3Problem> &hp = probs_host;
        for (auto &g : groups) {
            g.map_capacity = 64 + 4 * (int64_t)std::max(small_below, 256) + 1024;  // one M tile per block = most blocks any split can produce (+ the small-tile mode)
            for (int ci : g.conv_idx) {
                const ConvOp &cv = convs[ci];
                const Tensor &to = tensors[cv.out_t];
                const int bm = conv3x3_lds_bm(cv.ks, cv.nr, to.wp);
                g.map_capacity += (int64_t)((max_batch * to.hpwp + bm - 1) / bm) * cv.ntiles;
            }
            for (MapSlot &sl : g.slot) {
                if (!alloc_slot(sl, g.map_capacity, 0)) return false;
                workspace_bytes += g.map_capacity * (int64_t)sizeof(int2);
                if (queue_on && dtype == HRN_BF16) {
                    if (!hip_ok(hipMalloc((void **)&sl.q_dev, (size_t)g.map_capacity * sizeof(QUnit)), "hipMalloc(queue units)")) return false;
                    if (!hip_ok(hipHostMalloc((void **)&sl.q_pin, (size_t)g.map_capacity * sizeof(QUnit), hipHostMallocDefault), "hipHostMalloc(queue units)"))
                        return false;
                    workspace_bytes += g.map_capacity * (int64_t)sizeof(QUnit);
                }
            }
        }
        if (queue_on && dtype == HRN_BF16) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) num_cus = prop.multiProcessorCount;
            if (!hip_ok(hipMalloc((void **)&qheads_dev, groups.size() * 16 * sizeof(int)), "hipMalloc(queue heads)")) return false;
        }
        if (!hip_ok(hipMalloc((void **)&probs_dev, nprob * sizeof(Conv3Problem)), "hipMalloc(problems)")) return false;
        return hip_ok(hipMemcpy(probs_dev, hp.data(), nprob * sizeof(Conv3Problem), hipMemcpyHostToDevice),
                      "hipMemcpy(problems)");
    }

    void free_all() {
        if (plan_only) {
            free(blob);
        } else {
            (void)hipSetDevice(device);
            for (auto &b : buffers)
                if (b.dev) (void)hipFree(b.dev);
            if (blob) (void)hipFree(blob);
            if (tta_hm) (void)hipFree(tta_hm);
            if (pre_tmp) (void)hipFree(pre_tmp);
            if (rs_taps) (void)hipFree(rs_taps);
            rs_taps = nullptr, rs_taps_cap = 0;
            if (rs_done) (void)hipEventDestroy(rs_done);
            rs_done = nullptr;
            if (pass_done) (void)hipEventDestroy(pass_done);
            pass_done = nullptr;
            if (pre_params) (void)hipFree(pre_params);
            if (part_val) (void)hipFree(part_val);
            if (part_idx) (void)hipFree(part_idx);
            if (probs_dev) (void)hipFree(probs_dev);
            if (qheads_dev) (void)hipFree(qheads_dev);
            if (s2probs_dev) (void)hipFree(s2probs_dev);
            for (auto &g : s2groups)
                for (MapSlot &sl : g.slot) free_slot(sl);
            for (auto &g : groups)
                for (MapSlot &sl : g.slot) free_slot(sl);
            for (auto &g : dgroups)
                for (MapSlot &sl : g.slot) free_slot(sl);
            for (int k = 0; k < kPreRing; ++k) {
                if (pre_pin[k]) (void)hipHostFree(pre_pin[k]);
                if (pre_landed[k]) (void)hipEventDestroy(pre_landed[k]);
                pre_pin[k] = nullptr, pre_landed[k] = nullptr, pre_pin_bytes[k] = 0;
            }
        }
        blob = nullptr;
    }

    char *row0(int t) const {
        const Tensor &tt = tensors[t];
        const Buffer &b = buffers[tt.buf];
        return b.dev ? b.dev + b.lead_rows * tt.c * esize : nullptr;
    }

    // ---------------------------------------------------------------- weights
    struct Src {
        const float *p = nullptr;
        int64_t count = 0;
    };

    bool lookup(const std::map<std::string, Src> &m, const std::string &key, int64_t expect, const float **out) {
        auto it = m.find(key);
        if (it == m.end()) {
            err = "state_dict is missing key '" + key + "'";
            return false;
        }
        if (it->second.count != expect) {
            err = "state_dict key '" + key + "' has " + std::to_string(it->second.count) + " elements, expected " +
                  std::to_string(expect);
            return false;
        }
        *out = it->second.p;
        return true;
    }

    // scale/shift of an eval-mode BatchNorm2d (eps 1e-5): y = x*scale + shift
    bool bn_fold(const std::map<std::string, Src> &m, const std::string &bn, int ch, std::vector<double> &scale,
                 std::vector<double> &shift) {
        const float *g, *b, *mu, *var;
        if (!lookup(m, bn + ".weight", ch, &g) || !lookup(m, bn + ".bias", ch, &b) ||
            !lookup(m, bn + ".running_mean", ch, &mu) || !lookup(m, bn + ".running_var", ch, &var))
            return false;
        scale.resize(ch), shift.resize(ch);
        for (int i = 0; i < ch; ++i) {
            scale[i] = (double)g[i] / std::sqrt((double)var[i] + 1e-5);
            shift[i] = (double)b[i] - (double)mu[i] * scale[i];
        }
        return true;
    }

    bool load_weights(const hrn_tensor_desc *descs, int n) {
        std::map<std::string, Src> m;
        for (int i = 0; i < n; ++i) {
            if (descs[i].dtype != HRN_T_F32 || !descs[i].data || !descs[i].name) continue;
            int64_t cnt = 1;
            for (int d = 0; d < descs[i].ndim; ++d) cnt *= descs[i].dims[d];
            m[descs[i].name] = Src{(const float *)descs[i].data, cnt};
        }
        std::vector<char> host((size_t)blob_bytes, 0);
        std::vector<double> scale, shift;
        if (model == 1) {  // PoseResNet stem: w[k = (ci*7 + kh)*7 + kw][co]
            const float *w;
            if (!lookup(m, "conv1.weight", 64 * 147, &w) || !bn_fold(m, "bn1", 64, scale, shift)) return false;
            float *dw = (float *)(host.data() + stem_w_off), *db = (float *)(host.data() + stem_b_off);
            for (int co = 0; co < 64; ++co) {
                for (int k = 0; k < 147; ++k) dw[k * 64 + co] = (float)((double)w[co * 147 + k] * scale[co]);
                db[co] = (float)shift[co];
            }
            // MFMA image: five K chunks (k = (ci*7 + kh)*7 + kw, zero for k >= 147), 4 fragments, cout permutation NR = 4
            uint16_t *dp = (uint16_t *)(host.data() + stem_wp_off);
            for (int kc = 0; kc < 5; ++kc)
                for (int j = 0; j < 4; ++j)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int li = lane & 15, g = lane >> 4;
                        const int co = (li >> 2) * 16 + j * 4 + (li & 3);
                        for (int e = 0; e < 8; ++e) {
                            const int k = kc * 32 + g * 8 + e;
                            dp[((kc * 4 + j) * 64 + lane) * 8 + e] =
                                f32_to_bf16_host(k < 147 ? (float)((double)w[co * 147 + k] * scale[co]) : 0.f);
                        }
                    }
        } else {  // stem: w[k = ci*9+kh*3+kw][co]
            const float *w;
            if (!lookup(m, "conv1.weight", 64 * 27, &w) || !bn_fold(m, "bn1", 64, scale, shift)) return false;
            float *dw = (float *)(host.data() + stem_w_off), *db = (float *)(host.data() + stem_b_off);
            for (int co = 0; co < 64; ++co) {
                for (int k = 0; k < 27; ++k) dw[k * 64 + co] = (float)((double)w[co * 27 + k] * scale[co]);
                db[co] = (float)shift[co];
            }
            // MFMA image: one K chunk (k = ci*9 + kh*3 + kw, zero for k >= 27), 4 fragments, the usual cout
            // permutation with NR = 4 (lane owns 16 contiguous channels)
            uint16_t *dp = (uint16_t *)(host.data() + stem_wp_off);
            for (int j = 0; j < 4; ++j)
                for (int lane = 0; lane < 64; ++lane) {
                    const int li = lane & 15, g = lane >> 4;
                    const int co = (li >> 2) * 16 + j * 4 + (li & 3);
                    for (int e = 0; e < 8; ++e) {
                        const int k = g * 8 + e;
                        dp[(j * 64 + lane) * 8 + e] = f32_to_bf16_host(k < 27 ? (float)((double)w[co * 27 + k] * scale[co]) : 0.f);
                    }
                }
        }
        std::vector<float> wf;
        for (auto &cv : convs) {
            const float *w;
            const int kk = cv.k * cv.k, K = kk * cv.cin;
            if (cv.up) {
                // phase (a, b) of ConvTranspose2d(cin, cout, 4, stride 2, padding 1), weight [ci][co][ky][kx]:
                // out(2y + a, 2x + b) = sum over input (y + dy, x + dx) with ky = a + 1 - 2*dy in [0, 4), i.e.
                // a = 0: (dy 0, ky 1), (dy -1, ky 3);  a = 1: (dy +1, ky 0), (dy 0, ky 2) -- the same along x.
                if (!lookup(m, cv.conv + ".weight", (int64_t)cv.cin * cv.cout * 16, &w) ||
                    !bn_fold(m, cv.bn, cv.cout, scale, shift))
                    return false;
                const int a = (cv.up - 1) >> 1, b = (cv.up - 1) & 1;
                wf.assign((size_t)cv.cout * K, 0.f);   // K = 4*cin: live tap t = ty*2 + tx, dy = ty - 1 + a, dx = tx - 1 + b
                for (int t = 0; t < 4; ++t) {
                    const int dy = (t >> 1) - 1 + a, dx = (t & 1) - 1 + b;
                    const int ky = a + 1 - 2 * dy, kx = b + 1 - 2 * dx;   // in [0, 4) by construction
                    for (int co = 0; co < cv.cout; ++co)
                        for (int ci = 0; ci < cv.cin; ++ci)
                            wf[(size_t)co * K + t * cv.cin + ci] =
                                (float)((double)w[(((size_t)ci * cv.cout + co) * 4 + ky) * 4 + kx] * scale[co]);
                }
                pack_conv(cv, wf.data(), K, host.data() + cv.w_off);
                float *db = (float *)(host.data() + cv.b_off);
                for (int co = 0; co < cv.cout; ++co) db[co] = (float)shift[co];
                continue;
            }
            if (!lookup(m, cv.conv + ".weight", (int64_t)cv.cout * K, &w) ||
                !bn_fold(m, cv.bn, cv.cout, scale, shift))
                return false;
            // fold + reorder OIHW -> [co][tap*cin + ci]
            wf.assign((size_t)cv.cout * K, 0.f);
            for (int co = 0; co < cv.cout; ++co)
                for (int ci = 0; ci < cv.cin; ++ci)
                    for (int t = 0; t < kk; ++t)
                        wf[(size_t)co * K + t * cv.cin + ci] =
                            (float)((double)w[((size_t)co * cv.cin + ci) * kk + t] * scale[co]);
            if (cv.algo == 1)
                pack_conv_lds(cv, wf.data(), K, host.data() + cv.w_off);
            else
                pack_conv(cv, wf.data(), K, host.data() + cv.w_off);
            if (cv.s2) {
                ConvOp img = cv;   // same folded weights as ONE slice of all cin channels: k = tap * cin + ci, as the generic kernel orders it
                img.ks = cv.cin, img.nr = s2_frags_per_part(cv.cin), img.slices = 1, img.ntiles = cv.cout / (16 * img.nr);
                img.nch = (9 * cv.cin + 31) / 32, img.n96 = false;
                pack_conv_lds(img, wf.data(), K, host.data() + cv.w2_off);
            }
            float *db = (float *)(host.data() + cv.b_off);
            for (int co = 0; co < cv.cout; ++co) db[co] = (float)shift[co];
        }
        {
            const float *w, *b;
            const int c = head_c;  // channels final_layer reads
            if (!lookup(m, "final_layer.weight", (int64_t)joints * c, &w) || !lookup(m, "final_layer.bias", joints, &b))
                return false;
            memcpy(host.data() + head_w_off, w, sizeof(float) * joints * c);
            memcpy(host.data() + head_b_off, b, sizeof(float) * joints);
            // MFMA image: fragment f, chunk kc, lane (li, g): joint f*16 + li, k = kc*32 + g*8 + e
            const int kch = (c + 31) / 32;
            uint16_t *img = (uint16_t *)(host.data() + head_wp_off);
            for (int f = 0; f < 2; ++f)
                for (int kc = 0; kc < kch; ++kc)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int j = f * 16 + (lane & 15), k = kc * 32 + (lane >> 4) * 8 + e;
                            img[((size_t)(f * kch + kc) * 64 + lane) * 8 + e] =
                                (j < joints && k < c) ? f32_to_bf16_host(w[(size_t)j * c + k]) : (uint16_t)0;
                        }
        }
        if (plan_only) {
            memcpy(blob, host.data(), (size_t)blob_bytes);
        } else {
            if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return false;
            if (!hip_ok(hipMemcpy(blob, host.data(), (size_t)blob_bytes, hipMemcpyHostToDevice), "hipMemcpy(weights)"))
                return false;
        }
        weights_loaded = true;
        return true;
    }

    // Fragment-major packing.  One fragment = 16 packed rows x one K-chunk = 64 lanes x 16 B,
    // stored lane-linear so a wave loads it with one coalesced 1 KiB access.  Fragment f = ng*NR + j holds,
    // in packed row i (= lane & 15), output channel  ng*16*NR + (i>>2)*4*NR + j*4 + (i&3); lane group
    // g = lane>>4 holds k = kc*KC + g*VEC + [0,VEC).  With the operand swap D = W * X^T each lane then owns
    // 4*NR contiguous channels of one pixel.
    void pack_conv(const ConvOp &cv, const float *wf, int K, char *dst) const {
        const int KC = dtype == HRN_BF16 ? 32 : 16, VEC = dtype == HRN_BF16 ? 8 : 4;
        const int nfrag = cv.cout / 16;
        for (int f = 0; f < nfrag; ++f) {
            const int ng = f / cv.nr, j = f % cv.nr;
            for (int kc = 0; kc < cv.kchunks; ++kc)
                for (int lane = 0; lane < 64; ++lane) {
                    const int li = lane & 15, g = lane >> 4;
                    const int co = ng * 16 * cv.nr + (li >> 2) * 4 * cv.nr + j * 4 + (li & 3);
                    char *d = dst + (((size_t)f * cv.kchunks + kc) * 64 + lane) * 16;
                    for (int e = 0; e < VEC; ++e) {
                        const int k = kc * KC + g * VEC + e;
                        const float v = k < K ? wf[(size_t)co * K + k] : 0.f;
                        if (dtype == HRN_BF16)
                            ((uint16_t *)d)[e] = f32_to_bf16_host(v);
                        else
                            ((float *)d)[e] = v;
                    }
                }
        }
    }

    // Slice-major image for conv3x3_lds_kernel: block (cout tile t, slice s) is the exact LDS image
    // [chunk c][frag j][lane][8 bf16]; within a slice k = tap*KS + ci_local, zero beyond 9*KS.
    void pack_conv_lds(const ConvOp &cv, const float *wf, int K, char *dst) const {
        const int KS = cv.ks, NRB = cv.nr;
        if (KS == 16) {  // fp32 form (conv3x3_f32.hip): [cout tile][slice][9 taps][frag][lane][4 fp32], k = 4 g + e of the slice's 16
            for (int t = 0; t < cv.ntiles; ++t)
                for (int s = 0; s < cv.slices; ++s) {
                    float *blk = (float *)(dst + ((size_t)t * cv.slices + s) * cv.nch * NRB * 1024);
                    for (int c = 0; c < 9; ++c)
                        for (int j = 0; j < NRB; ++j)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int li = lane & 15, g = lane >> 4;
                                const int co = t * 16 * NRB + (li >> 2) * 4 * NRB + j * 4 + (li & 3);
                                for (int e = 0; e < 4; ++e)
                                    blk[((size_t)(c * NRB + j) * 64 + lane) * 4 + e] = wf[(size_t)co * K + c * cv.cin + s * 16 + 4 * g + e];
                            }
                }
            return;
        }
        for (int t = 0; t < cv.ntiles; ++t)
            for (int s = 0; s < cv.slices; ++s) {
                uint16_t *blk = (uint16_t *)(dst + ((size_t)t * cv.slices + s) * cv.nch * NRB * 1024);
                for (int c = 0; c < cv.nch; ++c)
                    for (int j = 0; j < NRB; ++j)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int li = lane & 15, g = lane >> 4;
                            // (96-cout form: conv3x3_n96.inc N96_CH64 -- a lane's channels are 8 contiguous ones per 32-channel group)
                            const int co = (cv.n96 && conv3x3_n96_ch64()) ? t * 96 + (j >> 1) * 32 + (li >> 2) * 8 + (j & 1) * 4 + (li & 3)
                                                                          : t * 16 * NRB + (li >> 2) * 4 * NRB + j * 4 + (li & 3);
                            uint16_t *d = blk + ((size_t)(c * NRB + j) * 64 + lane) * 8;
                            for (int e = 0; e < 8; ++e) {
                                const int kl = 32 * c + 8 * g + e;
                                float v = 0.f;
                                if (kl < 9 * KS) {
                                    const int tap = kl / KS, cil = kl % KS;
                                    v = wf[(size_t)co * K + tap * cv.cin + s * KS + cil];
                                }
                                d[e] = f32_to_bf16_host(v);
                            }
                        }
            }
    }

    // ---------------------------------------------------------------- execution
    struct Timing {
        std::vector<hipEvent_t> ev;  // ops.size()+1 events
    };

    bool run_pass(const float *images, int nb, const void *boxes, int box_dtype, float *pts, float *heatmaps,
                  hipStream_t s, Timing *tm, int flip = 0, const TapReq *tap = nullptr) {
        if (tm && !hip_ok(hipEventRecord(tm->ev[0], s), "hipEventRecord")) return false;
        // the list heads of every work-queue launch of the pass, in one go (each launch has its own eight)
        if (qheads_dev && !hip_ok(hipMemsetAsync(qheads_dev, 0, groups.size() * 16 * sizeof(int), s), "hipMemsetAsync(queue heads)")) return false;
        for (size_t oi = 0; oi < ops.size(); ++oi) {
            // every other launch walks its tensors backwards: it starts on what its producer wrote last, i.e. on the
            // part most likely still in the Infinity Cache (256 MB; a 256-crop tensor is up to 0.9 GB)
            const bool rev = alternate && (oi & 1);
            const hipError_t e = exec_op(ops[oi], rev, images, nb, boxes, box_dtype, pts, heatmaps, s, flip);
            if (!hip_ok(e, "kernel launch")) return false;
            if (tap && tap->op == (int)oi) {  // debug tap: the tensor this launch completed, as (ncrops, C, H, W) fp32
                const Tensor &t = tensors[tap->tensor];
                TapArgs a;
                a.in = row0(tap->tensor), a.dst = tap->dst;
                a.c = t.c, a.h = t.h, a.w = t.w, a.wp = t.wp, a.hpwp = t.hpwp;
                a.crop0 = tap->crop0, a.ncrops = tap->ncrops, a.crop_step = tap->crop_step;
                if (!hip_ok(launch_tap(dtype, a, s), "tap launch")) return false;
            }
            if (tm && !hip_ok(hipEventRecord(tm->ev[oi + 1], s), "hipEventRecord")) return false;
        }
        return true;
    }

    hipError_t exec_op(const Op &op, bool rev, const float *images, int nb, const void *boxes, int box_dtype, float *pts,
                       float *heatmaps, hipStream_t s, int flip) {
        hipError_t e = hipSuccess;
        switch (op.kind) {
        case OP_STEM: {
            const Tensor &t = tensors[stem_out_t];
            StemArgs a;
            a.images = images, a.out = row0(stem_out_t);
            a.w = (const float *)(blob + stem_w_off), a.bias = (const float *)(blob + stem_b_off);
            a.wp = (dtype == HRN_BF16 && !disable_stem_mfma) ? (const void *)(blob + stem_wp_off) : nullptr;
            a.n = nb, a.H = H, a.W = W;
            a.out_h = t.h, a.out_w = t.w, a.out_wp = t.wp, a.out_hpwp = t.hpwp;
            a.flip = flip;
            e = launch_stem(dtype, a, s);
            break;
        }
        case OP_STEM7: {
            const Tensor &t = tensors[stem_out_t];
            Stem7Args a;
            a.images = images, a.out = row0(stem_out_t);
            a.w = (const float *)(blob + stem_w_off), a.bias = (const float *)(blob + stem_b_off);
            a.wp = (dtype == HRN_BF16 && !disable_stem_mfma) ? (const void *)(blob + stem_wp_off) : nullptr;
            a.n = nb, a.H = H, a.W = W;
            a.out_h = t.h, a.out_w = t.w, a.out_wp = t.wp, a.out_hpwp = t.hpwp;
            a.flip = flip;
            e = launch_stem7(dtype, a, s);
            break;
        }
        case OP_MAXPOOL: {
            const Tensor &ti = tensors[pool_in_t], &to = tensors[pool_out_t];
            PoolArgs a;
            a.in = row0(pool_in_t), a.out = row0(pool_out_t);
            a.c = to.c, a.n = nb, a.in_wp = ti.wp, a.in_hpwp = ti.hpwp;
            a.out_h = to.h, a.out_w = to.w, a.out_wp = to.wp, a.out_hpwp = to.hpwp;
            e = launch_maxpool(dtype, a, s);
            break;
        }
        case OP_CONV: {
            const ConvOp &cv = convs[op.idx];
            const ConvArgs a = conv_args(cv, nb, rev);
            e = launch_conv(dtype, a, cv.nr, s);
            break;
        }
        case OP_CONV3_GROUP: {
            Conv3Group &g = groups[op.idx];
            bool hit;
            MapSlot *sl = find_slot(g.slot, nb, &hit);
            if ((e = slot_ready(sl, hit, s)) != hipSuccess) break;
            if (!hit) {  // block map depends on the micro-batch size: build it once per size (kMapSlots sizes kept)
                sl->q_units = -1;
                if (sl->q_dev && queue_plan(g, nb, rev, &g.units_host, &sl->q_bbf_prob, &sl->q_bbf_blocks, &sl->q_bbf_tiles) &&
                    (int64_t)g.units_host.size() <= g.map_capacity) {
                    sl->q_units = (int)g.units_host.size();
                    memcpy(sl->q_pin, g.units_host.data(), (size_t)sl->q_units * sizeof(QUnit));
                    e = hipMemcpyAsync(sl->q_dev, sl->q_pin, (size_t)sl->q_units * sizeof(QUnit), hipMemcpyHostToDevice, s);
                } else {
                    sl->nblocks = group_blocks(g, nb, &g.map_host, rev);
                    if (sl->nblocks > g.map_capacity) {   // (never write past the slot)
                        e = hipErrorInvalidValue;
                        break;
                    }
                    memcpy(sl->pin, g.map_host.data(), (size_t)sl->nblocks * sizeof(int2));
                    e = hipMemcpyAsync(sl->dev, sl->pin, (size_t)sl->nblocks * sizeof(int2), hipMemcpyHostToDevice, s);
                }
                if (e == hipSuccess) e = hipEventRecord(sl->landed, s);
                if (e != hipSuccess) break;
                sl->nb = nb;
                ++map_builds;
            }
            if (sl->q_units >= 0)
                e = launch_conv3x3_queue(sl->q_dev, sl->q_units, qheads_dev + 16 * op.idx, probs_dev + g.prob_first, sl->q_bbf_prob, sl->q_bbf_blocks,
                                         sl->q_bbf_tiles, rev ? 1 : 0, nb, num_cus, s);
            else
                e = launch_conv3x3_lds(probs_dev + g.prob_first, sl->dev, sl->nblocks, nb, convs[g.conv_idx[0]].ks,
                                       convs[g.conv_idx[0]].nr, s);
            break;
        }
        case OP_CONV_GROUP: {
            DirectGroup &g = dgroups[op.idx];
            bool hit;
            MapSlot *sl = find_slot(g.slot, nb, &hit);
            if ((e = slot_ready(sl, hit, s)) != hipSuccess) break;
            if (!hit) {  // descriptors (row counts) and block map depend on the micro-batch size
                for (size_t k = 0; k < g.conv_idx.size(); ++k) sl->args_pin[k] = conv_args(convs[g.conv_idx[k]], nb, rev);
                sl->mr = 4;  // shorter M tiles for small launches: fill the chip, shorten the serial K loop per block
                while (sl->mr > 1 && direct_group_blocks(g, nb, nullptr, sl->mr) < 512) sl->mr >>= 1;
                sl->nblocks = direct_group_blocks(g, nb, &g.map_host, sl->mr);
                memcpy(sl->pin, g.map_host.data(), (size_t)sl->nblocks * sizeof(int2));
                e = hipMemcpyAsync(sl->args_dev, sl->args_pin, g.conv_idx.size() * sizeof(ConvArgs), hipMemcpyHostToDevice, s);
                if (e == hipSuccess)
                    e = hipMemcpyAsync(sl->dev, sl->pin, (size_t)sl->nblocks * sizeof(int2), hipMemcpyHostToDevice, s);
                if (e == hipSuccess) e = hipEventRecord(sl->landed, s);
                if (e != hipSuccess) break;
                sl->nb = nb;
                ++map_builds;
            }
            e = launch_conv_group(dtype, sl->args_dev, sl->dev, sl->nblocks, g.nr, sl->mr, direct_wlds && dtype == 1, s);
            break;
        }
        case OP_S2_GROUP: {
            S2Group &g = s2groups[op.idx];
            if (!s2_active(g, nb)) {  // too few tiles for the slab kernel: the same convolutions on the generic kernel
                for (const Op &f : g.fallback) {
                    e = exec_op(f, rev, images, nb, boxes, box_dtype, pts, heatmaps, s, flip);
                    if (e != hipSuccess) break;
                }
                break;
            }
            bool hit;
            MapSlot *sl = find_slot(g.slot, nb, &hit);
            if ((e = slot_ready(sl, hit, s)) != hipSuccess) break;
            if (!hit) {
                sl->nblocks = s2_blocks(g, nb, &g.map_host);
                memcpy(sl->pin, g.map_host.data(), (size_t)sl->nblocks * sizeof(int2));
                e = hipMemcpyAsync(sl->dev, sl->pin, (size_t)sl->nblocks * sizeof(int2), hipMemcpyHostToDevice, s);
                if (e == hipSuccess) e = hipEventRecord(sl->landed, s);
                if (e != hipSuccess) break;
                sl->nb = nb;
                ++map_builds;
            }
            e = launch_conv_s2(s2probs_dev + g.prob_first, sl->dev, sl->nblocks, s);
            break;
        }
        case OP_CHAIN: {
            const Chain &ch = chains[op.idx];
            const ConvOp &c3 = convs[ch.conv3], &c1 = convs[ch.conv1];
            const Tensor &to = tensors[c3.out_t];
            ChainArgs a;
            a.in = row0(c3.in_t), a.res = row0(c3.res_t), a.out_y = row0(c3.out_t), a.out_t = row0(c1.out_t);
            a.w3 = blob + c3.w_off, a.b3 = (const float *)(blob + c3.b_off);
            a.w1 = blob + c1.w_off, a.b1 = (const float *)(blob + c1.b_off);
            a.x = nullptr, a.wds = nullptr, a.bds = nullptr;
            if (ch.ds >= 0) {
                const ConvOp &cd = convs[ch.ds];
                a.x = row0(cd.in_t), a.wds = blob + cd.w_off, a.bds = (const float *)(blob + cd.b_off);
                a.res = nullptr;
            }
            a.m = nb * to.hpwp, a.h = to.h, a.w = to.w, a.wp = to.wp, a.hpwp = to.hpwp, a.rev = rev;
            e = launch_bottleneck_chain(a, s);
            break;
        }
        case OP_FUSE: {
            FuseGroupArgs ga;
            ga.nf = fuses[op.idx].group;
            for (int k = 0; k < ga.nf; ++k) {
                const FuseOp &f = fuses[op.idx + k];
                const Tensor &to = tensors[f.out_t];
                FuseArgs &a = ga.f[k];
                a.nterms = f.nterms;
                for (int i = 0; i < f.nterms; ++i) {
                    const Tensor &tt = tensors[f.term_t[i]];
                    a.t[i].ptr = row0(f.term_t[i]), a.t[i].shift = f.shift[i];
                    a.t[i].wp = tt.wp, a.t[i].hpwp = tt.hpwp;
                }
                for (int i = f.nterms; i < 4; ++i) a.t[i] = FuseTerm{nullptr, 0, 0, 0};
                a.out = row0(f.out_t), a.c = to.c, a.h = to.h, a.w = to.w, a.wp = to.wp, a.hpwp = to.hpwp;
                a.m = nb * to.hpwp, a.rev = rev;
            }
            e = ga.nf == 1 ? launch_fuse(dtype, ga.f[0], s) : launch_fuse_group(dtype, ga, s);
            break;
        }
        case OP_HEAD: {
            const Tensor &t = tensors[head_in_t];
            HeadArgs a;
            a.in = row0(head_in_t);
            a.wgt = (const float *)(blob + head_w_off), a.bias = (const float *)(blob + head_b_off);
            a.wimg = (dtype == HRN_BF16 && !disable_head_mfma) ? (const void *)(blob + head_wp_off) : nullptr;
            a.heatmaps = heatmaps, a.part_val = part_val, a.part_idx = part_idx;
            a.n = nb, a.c = t.c, a.joints = joints, a.h = t.h, a.w = t.w, a.wp = t.wp, a.hpwp = t.hpwp;
            a.slabs = head_slabs, a.slab_px = head_slab_px;
            e = launch_head(dtype, a, s);
            break;
        }
        case OP_DECODE: {
            if (!pts) break;
            const Tensor &t = tensors[head_in_t];
            DecodeArgs a;
            a.part_val = part_val, a.part_idx = part_idx, a.boxes = boxes;
            a.box_is_float = box_dtype == HRN_BOX_F32, a.pts = pts;
            a.n = nb, a.joints = joints, a.h = t.h, a.w = t.w, a.slabs = head_slabs;
            e = launch_decode(a, s);
            break;
        }
    }
        return e;
    }

    bool check_forward_args(const void *images, int n, const void *boxes, float *pts, float *heatmaps) {
        if (plan_only) {
            err = "plan-only handle (device_id < 0): there is no CPU compute path";
            return false;
        }
        if (!weights_loaded) {
            err = "weights not loaded (call hrn_load_weights or hrn_adopt_weights)";
            return false;
        }
        if (n < 0 || (n > 0 && !images)) {
            err = "bad images / n";
            return false;
        }
        if (!pts && !heatmaps) {
            err = "both pts and heatmaps are NULL";
            return false;
        }
        if (pts && !boxes && n > 0) {
            err = "pts requested without boxes";
            return false;
        }
        return true;
    }
};

// ====================================================================================================
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
    std::unique_ptr<hrn_ctx> h(new hrn_ctx());
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
    if (n > 0 && !h->pass_enter((hipStream_t)stream)) return 6;
    for (int off = 0; off < n; off += h->max_batch) {
        const int nb = n - off < h->max_batch ? n - off : h->max_batch;
        const float *img = (const float *)images_dev + (size_t)off * 3 * h->H * h->W;
        const void *bx = boxes_dev ? (const char *)boxes_dev + (size_t)off * 16 : nullptr;
        float *p = pts_dev ? pts_dev + (size_t)off * h->joints * 3 : nullptr;
        float *hp = heatmaps_dev ? heatmaps_dev + (size_t)off * h->joints * hm : nullptr;
        if (!h->run_pass(img, nb, bx, box_dtype, p, hp, (hipStream_t)stream, nullptr)) return 8;
    }
    if (n > 0 && !h->pass_leave((hipStream_t)stream)) return 6;
    return 0;
}

// Flip test-time augmentation + the evaluation decode (testing/Test.py:132-140, training/COCO.py:206-230,
// misc/utils.py:9-29, 125-175): two passes per micro-batch (the second reads the crops mirrored in the stem), then one
// kernel averages, finds the maxima and applies the quarter-pixel refinement.
int hrn_forward_flip_tta(hrn_handle h, const void *images_dev, int n, const int32_t *flip_pairs_host, int npairs,
                         int post_processing, float *heatmaps_dev, float *preds_dev, float *maxvals_dev, void *stream) {
    if (!h) return 1;
    if (!h->check_forward_args(images_dev, n, nullptr, nullptr, heatmaps_dev ? heatmaps_dev : (float *)nullptr)) return 7;
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
    if (!h->pass_enter(s)) return 6;
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
    return h->pass_leave(s) ? 0 : 6;
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
    if (!h->check_forward_args(images_dev, n, nullptr, nullptr, (float *)1)) return 7;
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
    if (!h->pass_enter((hipStream_t)stream)) return 6;
    if (!h->run_pass((const float *)images_dev, n, nullptr, 0, nullptr, heatmaps_dev, (hipStream_t)stream, nullptr, 0, &req)) return 8;
    return h->pass_leave((hipStream_t)stream) ? 0 : 6;
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
int hrn_launches_per_pass(hrn_handle h) { return h ? (int)h->ops.size() : 0; }

int hrn_plan_queue(hrn_handle h, int group, int n, int reverse, int32_t *units, int capacity, int32_t *info) {
    if (!h || n <= 0 || n > h->max_batch) return -1;
    if (group < 0 || group >= (int)h->groups.size()) return -1;
    std::vector<QUnit> u;
    std::vector<int> uc;
    int bbf_prob = 0, bbf_blocks = 0, bbf_tiles = 0, bbf_conv = -1;
    if (!h->queue_plan(h->groups[group], n, reverse != 0, &u, &bbf_prob, &bbf_blocks, &bbf_tiles, &uc, &bbf_conv)) return -2;
    for (size_t i = 0; i < u.size() && (int)i < capacity; ++i) {
        int32_t *o = units + i * 4;
        o[0] = uc[i], o[1] = u[i].ch_base / 96, o[2] = u[i].mt0, o[3] = u[i].ntile;
    }
    if (info) info[0] = bbf_conv, info[1] = bbf_blocks, info[2] = bbf_tiles, info[3] = h->num_cus;
    return (int)u.size();
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
    if (group < 0 || group >= (int)h->s2groups.size()) return -1;
    const S2Group &g = h->s2groups[group];
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
    if (!h->check_forward_args(images_dev, n, nullptr, nullptr, (float *)1)) return 7;
    if (!h->hip_ok(hipSetDevice(h->device), "hipSetDevice")) return 6;
    hrn_ctx::Timing tm;
    tm.ev.resize(h->ops.size() + 1);
    for (auto &e : tm.ev)
        if (!h->hip_ok(hipEventCreate(&e), "hipEventCreate")) return 6;
    bool ok = h->pass_enter((hipStream_t)stream) &&
              h->run_pass((const float *)images_dev, n, nullptr, 0, nullptr, nullptr, (hipStream_t)stream, &tm) && h->pass_leave((hipStream_t)stream);
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
            } else if (op.kind == OP_CHAIN) {  // two or three 1x1 convs of equal FLOPs
                const hrn_ctx::Chain &ch = h->chains[op.idx];
                const float share = ch.ds >= 0 ? ms / 3.f : ms * 0.5f;
                if (conv_ms && ch.conv3 < conv_ms_len) conv_ms[ch.conv3] = share;
                if (conv_ms && ch.conv1 < conv_ms_len) conv_ms[ch.conv1] = share;
                if (conv_ms && ch.ds >= 0 && ch.ds < conv_ms_len) conv_ms[ch.ds] = share;
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
