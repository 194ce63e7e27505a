// Host-side types of the HRNet engine (hrnet_mi355.cpp): tensors and buffers of the flat padded NHWC workspace, one record per
// convolution of the compiled graph, the launch groups and their per-batch-size block maps.  Included inside hrnet_mi355.cpp's
// anonymous namespace.
#pragma once

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
    bool compact = false;    // 96-cout form with the compact enumeration of M (tiles of real pixels only: conv3x3_n96.inc)
    bool s2 = false;
    bool w2_image = false;   // the slab-kernel weight image exists although the conv itself is not `s2` (conv2 under the fused stem)
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

