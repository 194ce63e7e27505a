// Device-side interface of the HRNet hot path: argument blocks + launchers.
// Everything here is gfx950 (CDNA4) only; see DESIGN.md for layouts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdlib>
#include <cstring>

namespace hrn {

// Kernels that use more than 64 KiB of dynamic LDS: hipFuncAttributeMaxDynamicSharedMemorySize is a property of the function ON
// ONE DEVICE, so it is set once per (kernel, device) -- a process-wide "done" flag left every device but the first handle's without
// it (one process driving several GPUs: native.MultiDeviceHRNet).  `done` = one bit per device ordinal, a static of the caller.
inline hipError_t set_dynamic_lds(const void *func, int bytes, std::atomic<unsigned long long> &done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

// The HRN_* switches exist for same-box A/B runs and the bit-identity tests.  In production NONE of them is
// read: the library's behaviour does not depend on the caller's environment.  A process that sets HRN_DEBUG_ENV=1 (the test suite,
// tools/ab.sh) opts in; the switches are then read ONCE, when a handle is created (hrn_create snapshots what it saw:
// hrn_switches()), never during a call.
inline const char *hrn_env(const char *name) {
    static const bool debug = [] {
        const char *v = getenv("HRN_DEBUG_ENV");
        return v && *v && strcmp(v, "0") != 0;
    }();
    return debug ? getenv(name) : nullptr;
}

enum { DT_F32 = 0, DT_BF16 = 1 };

// Activation tensors use the "flat padded NHWC" layout:
//   row(n, r, c) = n*Hp*Wp + r*Wp + c,  Wp = W+1, Hp = H+1, C channels per row,
//   the extra column / row of every image hold zeros, as do the guard rows before
//   image 0 and after the last image.  A 3x3/pad-1 tap (dh,dw) is then the constant
//   row shift dh*Wp+dw -- no boundary tests in the inner loop.
struct ConvArgs {
    const void *in;     // row 0 of the input tensor
    void *out;          // row 0 of the output tensor
    const void *w;      // packed weights (fragment-major, see pack_conv_weights)
    const float *bias;  // folded BN bias, fp32[cout]
    const void *res;    // residual tensor (same geometry as out) or nullptr
    int cin, cout;
    int in_wp, in_hpwp;                     // input row pitch / image pitch (rows)
    int out_h, out_w, out_wp, out_hpwp;     // output geometry
    int m;                                  // rows to produce = n * out_hpwp
    int ksize, stride, relu;
    int kchunks;                            // padded K / (32 bf16 | 16 f32)
    int rev;                                // walk the M tiles backwards (see hrn_ctx::alternate)
    int pre_mode;                           // K = 64 1x1 convs with a residual: prefetch variant (1: residual before the K loop, 4-fragment tiles)
    int wlds;                               // bf16, full-size tiles: weights staged through LDS once per block (kernels.hip: WL)
    int xlds;                               // ... and, for the stride-2 3x3 convolutions with cin % 32 == 0, the pixel fragments through a per-wave LDS ring (kernels.hip: XL)
    // one phase (a, b) of a ConvTranspose2d(4, stride 2, padding 1) run as a 3x3 conv on the input grid: the result of
    // pixel (ho, wo) is stored at (2*ho + a, 2*wo + b) of the twice-as-large tensor (poseresnet.py:84-100)
    int up, up_a, up_b, up_wp, up_hpwp;
    int taps[4];                            // ksize == 2: row shift of the four live taps (tap-major K = 4*cin)
};

// LDS-staged 3x3 stride-1 convolution (conv3x3_lds.hip); input and output share one geometry.
// One device-resident descriptor per convolution; a launch covers a group of them.
struct Conv3Problem {
    const void *in;
    void *out;
    const void *w;      // slice-major packed weights: [cout tile][slice][chunk][frag][lane][16 B]
    const float *bias;
    const void *res;
    int cin, cout;
    int h, wd, wp, hpwp;
    int relu;
    int slices;         // cin / KS
    int ntiles;         // cout / (16*NRB)
    int tiles_per_block;  // consecutive M tiles one block walks (weights stay in LDS when slices == 1)
    int bm;               // pixels per M tile: 512, or 256 when two 512-row slabs would not fit in LDS
    // x / d == (x * magic) >> shift for x < 2^27, magic = floor(2^shift / d) + 1, shift = 30 + ceil(log2 d)
    unsigned magic_hpwp, magic_wp;
    int shift_hpwp, shift_wp;
    // fused BasicBlock (conv3x3_lds.hip: bbf_run): this problem is conv1, w2 / bias2 are conv2's, `out` is the block's
    // output and `in` doubles as the residual
    const void *w2;
    const float *bias2;
    // 96-cout form (conv3x3_n96.inc): KS = 32, ntiles = cout / 96, bm = 512 or 384
    int n96;
    // ... with the COMPACT enumeration of M (round 4): tiles of real pixels only; hw = h * wd; fast divisions by hw and wd;
    // rows of a tile's slab (the longest run of flat rows a tile of `bm` / of 128 pixels spans, + 2 wp + 2)
    int compact, hw;
    unsigned magic_hw, magic_w;
    int shift_hw, shift_w;
    int slab_rows, slab_rows_small;
};
int conv3x3_n96_ch64();           // output-channel permutation of the 96-cout form's weight image (conv3x3_n96.inc)
int conv3x3_lds_bbf_ok(int wp);  // the fused BasicBlock kernel fits this row pitch
int conv3x3_n96_max_rows();   // rows (of 64 bytes) one slab buffer of the 96-cout form holds
int conv3x3_lds_bm(int ks, int nrb, int wp);  // (32, 6) = the 96-cout form; ks = 16: the fp32 kernel (conv3x3_f32.hip)
int conv3x3_f32_bm(int wp);
hipError_t launch_conv3x3_f32(const Conv3Problem *probs_dev, const void *blockmap_dev, int nblocks, int nb, int nrb, hipStream_t s);

// Stride-2 slab kernel (conv_s2.hip): every 3x3 / stride-2 convolution with 48 (HRNet-W32: 32 / 64) input channels that reads ONE tensor at
// one fuse level contributes "parts" (48 (32) output channels each) to one problem; a block stages the input slab of `rows`
// output rows once and its waves keep the parts' weights in registers.
constexpr int kS2SlabBytes = 79872;   // one slab buffer: 832 sub-slots of 96 bytes (two of them + the biases fill the 160 KiB)
constexpr int s2_frags_per_part(int cin) { return cin == 48 ? 3 : 2; }   // 16-cout fragments a wave keeps in registers
// a pixel's cin * 2 bytes are kept in LDS as sub-slots of 96 (cin = 48, 96) or 32 (cin = 32, 64) bytes, one region per sub-slot
// index; a region is a whole number of 1-KiB LDS-DMA pieces
constexpr int s2_subslot_bytes(int cin) { return cin % 48 == 0 ? 96 : 32; }
constexpr int s2_region_bytes(int cin) { return kS2SlabBytes / (cin * 2 / s2_subslot_bytes(cin)) / 1024 * 1024; }
constexpr int s2_slot_capacity(int cin) { return s2_region_bytes(cin) / s2_subslot_bytes(cin); }   // input pixels one slab buffer holds
// The slab keeps TWO virtual input rows (4 wop slots: both column parities of both rows) + `pad` empty slots per output row, so
// that the 16 pixels of a fragment that wraps an output row stay on disjoint LDS banks: the wrap jumps 3 wop + 1 + pad slots,
// and sub-slots of 96 (32) bytes are conflict-free for 16 lanes iff their slot numbers are consecutive mod 8 (round 4: the
// unpadded slab measured a 32 % LDS bank-conflict rate, profiles/round3_pmc_s2.txt).
constexpr int s2_pair_pad(int wop) { return (8 - (3 * wop) % 8) % 8; }
constexpr int s2_pair_pitch(int wop) { return 4 * wop + s2_pair_pad(wop); }   // slots per output row of the slab
constexpr int kS2MaxParts = 8;
struct S2Part {
    const void *w;      // [K chunks][frags][64 lanes][16 B]: the (cin, frags) image of pack_conv_lds for this cout tile (k = tap * cin + ci)
    const float *bias;  // the convolution's folded bias (indexed with ch0 + c)
    void *out;          // row 0 of the convolution's output tensor
    int cout, ch0;      // channels per row of that tensor, first channel of this part
    int relu, pad_;
};
struct S2Problem {
    const void *in;     // row 0 of the input tensor
    int cin;            // 32, 48 or 64
    int in_wp, in_hpwp;
    int ho, wo, wop, out_hpwp;   // output geometry (shared by all parts)
    int rows;                    // output rows per tile
    int tiles_per_image;         // ceil(ho / rows)
    int nparts;                  // cout groups (of 16 * s2_frags_per_part(cin) channels)
    // wave w works on part wave_part[w] (0xff: idle) and on the pixel fragments wave_f0[w], + wave_fs[w], ... of every tile
    unsigned char wave_part[8], wave_f0[8], wave_fs[8];
    unsigned magic_wop;          // x / wop == (x * magic) >> shift
    int shift_wop;
    S2Part part[kS2MaxParts];
};
hipError_t launch_conv_s2(const S2Problem *probs_dev, const void *map_dev, int nblocks, hipStream_t s);

// The stem as one kernel (stem_fused.hip, round 4, bf16): conv1 computed into the stride-2 slab's LDS layout, conv2 from there;
// one output row of conv2 per tile.  `probs_dev[.]` = conv2 as a slab-kernel problem with rows = 1 (its `in` is not read),
// `stem` = conv1's arguments (its `out` is not written).
constexpr int kStemFuseRegionBytes = 15360;   // one 16-channel region of a slab buffer: 480 sub-slots of 32 bytes
constexpr int kStemFusePatchBytes = 12544;    // one patch buffer: 7 crop rows x 3 colours x (W + 8) bf16 (W <= 288: the slab's limit)
int stem_fused_fits(int wop, int w_in);

// layer1: conv3 (+shortcut, ReLU) of one Bottleneck and conv1 (+ReLU) of the next in one pass (bottleneck_chain.hip)
struct ChainArgs {
    const void *in;        // conv2 output of block b: [rows][64]
    const void *res;       // shortcut of block b: [rows][256]
    void *out_y;           // block b output: [rows][256]
    void *out_t;           // conv1 output of block b+1: [rows][64]
    const void *w3;        // 256 x 64, generic fragment image with NR = 2
    const float *b3;
    const void *w1;        // 64 x 256, generic fragment image with NR = 4 (nullptr with w2: the layer's last block, no next conv1)
    const float *b1;
    const void *in3;       // round 5 (w2 != nullptr): conv1's output t1 [rows][64]; conv2 (3x3) is computed here, `in` is not read
    const void *w2;        // 64 x 576, generic fragment image with NR = 2 (k = tap * 64 + ci), or nullptr
    const float *b2;
    const void *x;         // block 0 only (wds != nullptr): the block input [rows][64]; the shortcut is Wds*x + bds
    const void *wds;       // 256 x 64, generic fragment image with NR = 2, or nullptr (shortcut read from `res`)
    const float *bds;
    int m, h, w, wp, hpwp; // rows to produce = n*hpwp; geometry for the pad mask
    int rev;
    int max_blocks;        // persistent blocks of the launch (512: two per CU)
};
hipError_t launch_bottleneck_chain(const ChainArgs &a, hipStream_t s);

struct StemArgs {          // conv1 3->64 3x3 s2 + BN + ReLU, NCHW fp32 in, flat padded out
    const float *images;   // (n,3,H,W)
    void *out;
    const float *w;        // [27][64] fp32, folded
    const float *bias;     // [64]
    const void *wp;        // bf16 mode: MFMA image of the weights, [4 frags][64 lanes][8 bf16], K = 27 padded to 32
    int n, H, W;           // input size
    int out_h, out_w, out_wp, out_hpwp;
    int flip;              // read the crops mirrored left-right (flip-TTA: misc/utils.py flip_tensor(image, dim=-1))
};

hipError_t launch_stem_fused(const S2Problem *probs_dev, const void *map_dev, int nblocks, const StemArgs &stem, hipStream_t s);

// Crop pre-path (prepath.hip): one person's slice of the frame, its zero padding and where its horizontal pass lives
struct CropParams {
    int x1, y1, w_crop, h_crop;   // slice of the frame that is actually read (numpy clamps the stop index)
    int pad_top, pad_left;        // zero rows / columns in front of it (np.pad)
    int h_pad, w_pad;             // size of the padded crop = input of the resize
    long long tmp_off;            // byte offset of this crop's [h_pad][W][3] uint8 intermediate
};
hipError_t launch_prepath(const unsigned char *frame_dev, int frame_w, const CropParams *crops_dev, int n, int max_h_pad,
                          unsigned char *tmp_dev, float *images_dev, int H, int W, hipStream_t s);
// single-person pre-path (prepath.hip): cv2.resize of whole frames; one entry per output column, then per output row
struct ResizeTaps {
    int ofs;       // first source index of the window (may lie outside: replicate border)
    short c[4];    // fixed-point coefficients, 11 bits
};
hipError_t launch_resize_frames(const unsigned char *frames_dev, int n, int src_h, int src_w, int interp, ResizeTaps *taps_dev,
                                float *images_dev, int H, int W, hipStream_t s);

struct Stem7Args {         // PoseResNet conv1: 3->64 7x7 s2 p3 + BN + ReLU, NCHW fp32 in, flat padded out (poseresnet.py:25-27)
    const float *images;
    void *out;
    const float *w;        // [147][64] fp32, folded, k = (ci*7 + kh)*7 + kw
    const float *bias;     // [64]
    const void *wp;        // bf16 mode: MFMA image [5 chunks][4 frags][64 lanes][8 bf16], K = 147 padded to 160
    int n, H, W;
    int out_h, out_w, out_wp, out_hpwp;
    int flip;
};
struct PoolArgs {          // MaxPool2d(3, stride 2, padding 1) on a post-ReLU tensor (poseresnet.py:28): zero pads == -inf pads
    const void *in;
    void *out;
    int c, n;
    int in_wp, in_hpwp;
    int out_h, out_w, out_wp, out_hpwp;
};
hipError_t launch_stem7(int dtype, const Stem7Args &a, hipStream_t s);
hipError_t launch_maxpool(int dtype, const PoolArgs &a, hipStream_t s);

struct FuseTerm {
    const void *ptr;
    int shift;      // nearest-upsample factor 2^shift (0 = same resolution)
    int wp, hpwp;   // geometry of the term tensor
};
struct FuseArgs {          // out = relu(sum_j term_j) in order j = 0..nterms-1
    FuseTerm t[4];
    int nterms;
    void *out;
    int c, h, w, wp, hpwp;
    int m;                 // n * hpwp
    int rev;               // walk the rows backwards
};

struct FuseGroupArgs {     // up to four independent fuse outputs in one launch; block b belongs to f[k] with block_end[k-1] <= b < block_end[k]
    FuseArgs f[4];
    int nf;
    int block_end[4];
};
hipError_t launch_fuse_group(int dtype, FuseGroupArgs &g, hipStream_t s);

struct HeadArgs {          // final 1x1 conv (+bias) and per-(crop, joint) partial arg-max
    const void *in;        // fused branch 0, flat padded, c channels
    const float *wgt;      // [joints][c] fp32
    const void *wimg;      // bf16 mode: MFMA image of the weights, [2 frags][ceil(c/32) chunks][64 lanes][8 bf16]
    const float *bias;     // [joints]
    float *heatmaps;       // (n,joints,h,w) fp32 NCHW or nullptr
    float *part_val;       // [n][joints][slabs]
    int *part_idx;
    int n, c, joints, h, w, wp, hpwp, slabs, slab_px;
};

struct DecodeArgs {        // SimpleHRNet.py:297-308
    const float *part_val;
    const int *part_idx;
    const void *boxes;     // (n,4) int32 or fp32
    int box_is_float;
    float *pts;            // (n,joints,3)
    int n, joints, h, w, slabs;
};

struct TtaArgs {           // flip-TTA combine + get_max_preds + quarter-pixel refinement (misc/utils.py:19-29, 125-175)
    float *hm;             // (n,joints,h,w): plain pass in, average out
    const float *hm_flipped;  // the mirrored crops' heat-maps
    float *preds;          // (n,joints,2): x, y in heat-map pixels
    float *maxvals;        // (n,joints)
    int pair[32];          // joint j of the mirrored output is joint pair[j] (flip_back)
    int n, joints, h, w, post_processing;
};
hipError_t launch_tta_decode(const TtaArgs &a, hipStream_t s);

struct TapArgs {           // debug tap: crops crop0, crop0 + crop_step, ... of a flat padded tensor -> (ncrops, c, h, w) fp32
    const void *in;
    float *dst;
    int c, h, w, wp, hpwp, crop0, ncrops, crop_step;
};
hipError_t launch_tap(int dtype, const TapArgs &a, hipStream_t s);

struct PadCheckArgs {      // debug: one workspace buffer (guard rows + nmax images of (h + 1) x (w + 1) rows of c channels)
    const void *buf;       // allocation start (NOT row 0 of image 0)
    unsigned long long *count;
    long rows, lead_rows;  // rows of the buffer, guard rows in front of image 0
    int c, h, w, wp, hpwp, nmax;
};
hipError_t launch_pad_check(int dtype, const PadCheckArgs &a, hipStream_t s);

hipError_t launch_conv(int dtype, const ConvArgs &a, int nr, hipStream_t s);
// grouped launch of the generic kernel: device-resident ConvArgs[], block map entries (prob | cout tile << 8, M tile)
hipError_t launch_conv_group(int dtype, const ConvArgs *probs_dev, const void *map_dev, int nblocks, int nr, int mr, int wlds,
                             hipStream_t s);
hipError_t launch_conv3x3_lds(const Conv3Problem *probs_dev, const void *blockmap_dev, int nblocks, int nb, int ks,
                              int nrb, hipStream_t s);
hipError_t launch_stem(int dtype, const StemArgs &a, hipStream_t s);
hipError_t launch_fuse(int dtype, const FuseArgs &a, hipStream_t s);
hipError_t launch_head(int dtype, const HeadArgs &a, hipStream_t s);
hipError_t launch_decode(const DecodeArgs &a, hipStream_t s);

// rows per block of the conv kernels: buffers keep this many guard rows after the last image
// -- and the longest slab a compact tile of the 96-cout form may read past the last image (a last tile that holds a single
// real pixel still stages its whole slab, up to conv3x3_n96_max_rows() = 700 rows; a flat tile reads at most 512 + halo)
constexpr int kConvBlockRows = 704;

}  // namespace hrn
