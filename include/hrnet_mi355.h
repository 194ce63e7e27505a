/*
 * hrnet_mi355.h -- C ABI of the MI355X-native HRNet pose-inference hot path.
 *
 * This is the drop-in boundary behind SimpleHRNet.predict() of stefanopini/simple-HRNet.
 * The reference has no native interface on this path: the seam is the Python attribute
 * `self.model`, invoked as `self.model(images)` (SimpleHRNet.py:284-294, 419-429) and already
 * swapped for a foreign engine by the reference itself (TRTModule, SimpleHRNet.py:143-147);
 * the closest native precedent is `void _nms(int*, int*, const float*, int, int, float, int)`
 * (misc/nms/gpu_nms.hpp:1).  Every entry point below names the reference behaviour it replaces.
 *
 * Conventions
 *   - plain C types only: raw pointers + sizes; no torch / HIP types in signatures
 *     (`stream` is a hipStream_t passed as void*, NULL = the default stream);
 *   - every function returns 0 on success, non-zero on failure; the message is
 *     available from hrn_last_error() (the reference raises Python exceptions:
 *     SimpleHRNet.py:107,114,139,210 -- the ctypes shim turns our codes back into them);
 *   - a handle is bound to one GPU; it is not thread-safe, distinct handles are;
 *   - all work is stream-ordered on `stream`; nothing allocates inside hrn_forward().  A handle has ONE workspace: calls
 *     on different streams are serialised on the device by the handle itself (a call on another stream than the previous
 *     one first waits for that one's last kernel) -- concurrency comes from several handles, not from several streams;
 */
#ifndef HRNET_MI355_H
#define HRNET_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hrn_ctx *hrn_handle;

enum { HRN_F32 = 0, HRN_BF16 = 1 };            /* arithmetic / activation storage type       */
enum { HRN_BOX_I32 = 0, HRN_BOX_F32 = 1 };     /* boxes dtype: SimpleHRNet.py:230 vs :223      */
enum { HRN_T_F32 = 0, HRN_T_I64 = 1 };         /* hrn_tensor_desc.dtype                        */

/* One state_dict entry, exactly as `torch.load(checkpoint)` yields it
 * (SimpleHRNet.py:117-121): name, shape, contiguous host data (conv weights OIHW fp32). */
typedef struct {
    const char *name;
    const void *data;
    int32_t ndim;
    int64_t dims[4];
    int32_t dtype;
} hrn_tensor_desc;

/* Static description of one convolution of the compiled graph (for tests / tooling). */
typedef struct {
    char name[96];        /* state_dict prefix of the conv, e.g. "stage3.1.branches.2.0.conv1" */
    int32_t cin, cout, ksize, stride, relu, has_residual;   /* ksize 2 = one phase of a ConvTranspose2d(4, s2, p1) */
    int32_t in_h, in_w, out_h, out_w;
    int32_t kpad;         /* K = ksize*ksize*cin rounded up to the MFMA K-chunk                 */
    int32_t nr;           /* 16-wide cout fragments per group (packing parameter)              */
    int32_t algo;         /* 0 generic MFMA kernel, 1 pipelined LDS-staged 3x3 stride-1 kernel, 2 = 1 as half of a fused BasicBlock (HRN_BBF=1), 3 = the 96-cout form of 1 (ks 32, nr 6), 4 = generic plan + the stride-2 slab kernel (conv_s2.hip) for calls large enough */
    int32_t ks;           /* algo 1: input channels per LDS slice (k = tap*ks + ci inside one)  */
    int64_t w_offset;     /* byte offset of the packed weights in the blob                     */
    int64_t w_bytes;
    int64_t b_offset;     /* byte offset of the folded fp32 bias                                */
    double flops;         /* 2*MAC per crop                                                    */
} hrn_conv_info;

/* Replaces `HRNet(c, nof_joints)` + `.to(device).eval()` (SimpleHRNet.py:110,141-142; graph of
 * models_/hrnet.py:75-155).  height/width = network input resolution (multiples of 32),
 * max_batch = largest micro-batch one internal pass will process (workspace is sized for it;
 * hrn_forward accepts any n and chunks internally like SimpleHRNet.py:285-294).
 * device_id >= 0: HIP device.  device_id < 0: plan-only handle (no GPU touched; graph,
 * folding and packing run on the host so the host logic is testable on a CPU-only box;
 * hrn_forward fails on such a handle -- there is NO CPU compute path). */
int hrn_create(hrn_handle *out, int c, int nof_joints, int height, int width, int dtype, int max_batch,
               int device_id);

/* The other model SimpleHRNet offers (model_name='PoseResNet', SimpleHRNet.py:111-112; models_/poseresnet.py:16-122):
 * c is then the ResNet size (50 / 101 / 152).  hrn_create(...) == hrn_create_model(HRN_MODEL_HRNET, ...).
 * Everything else of the interface is model independent. */
enum { HRN_MODEL_HRNET = 0, HRN_MODEL_POSERESNET = 1 };
int hrn_create_model(hrn_handle *out, int model, int c, int nof_joints, int height, int width, int dtype, int max_batch,
                     int device_id);
void hrn_destroy(hrn_handle h);
const char *hrn_last_error(hrn_handle h); /* h may be NULL: error of the last failed hrn_create */

/* Replaces `model.load_state_dict(checkpoint)` (SimpleHRNet.py:117-121).  Folds every
 * (conv, BatchNorm) pair (eps 1e-5), repacks to the MFMA fragment layout and uploads. */
int hrn_load_weights(hrn_handle h, const hrn_tensor_desc *descs, int n);

/* Multi-GPU weight distribution (replaces DataParallel's per-forward `replicate`,
 * SimpleHRNet.py:135): the packed blob is one contiguous device buffer that rank 0 fills via
 * hrn_load_weights and the other ranks receive by ONE RCCL broadcast, then adopt. */
int64_t hrn_weight_blob_bytes(hrn_handle h);
void *hrn_weight_blob_ptr(hrn_handle h);            /* device pointer (host pointer if plan-only) */
int hrn_adopt_weights(hrn_handle h);                /* mark an externally filled blob as loaded   */
int hrn_weight_blob_read(hrn_handle h, int64_t offset, void *dst_host, int64_t nbytes);

/* Replaces the model call + decode loop, SimpleHRNet.py:281-308 (dup. 416-443):
 *   images_dev  (n,3,H,W) fp32 NCHW device pointer -- what SimpleHRNet hands to self.model
 *   boxes_dev   (n,4) [x1,y1,x2,y2] device pointer, int32 or fp32 per box_dtype; may be NULL
 *               when pts_dev is NULL
 *   pts_dev     (n,joints,3) fp32 device pointer (y, x, confidence) or NULL
 *   heatmaps_dev (n,joints,H/4,W/4) fp32 NCHW device pointer or NULL (return_heatmaps,
 *               and the level-1 "self.model(images)" seam)
 * At least one of pts_dev / heatmaps_dev must be non-NULL. */
int hrn_forward(hrn_handle h, const void *images_dev, int n, const void *boxes_dev, int box_dtype, float *pts_dev,
                float *heatmaps_dev, void *stream);

/* Crop pre-path of the single-image multi-person branch (SimpleHRNet.py:236-278): for each
 * detector box -- round, correct the aspect ratio by padding (:243-272), slice the BGR frame as RGB (:274), zero-pad
 * (:276), Resize((H,W)) with Pillow's antialiased bilinear filter, ToTensor, Normalize (:167-172) -- written straight
 * into the (n,3,H,W) fp32 batch hrn_forward reads.  Bit-identical to the reference's transform.
 *   frame_dev   (frame_h, frame_w, 3) uint8 BGR, device
 *   dets_host   (n, det_stride) float32 on the HOST, columns 0..3 = x1,y1,x2,y2 as the detector returns them
 *   images_dev  out: (n,3,H,W) float32, device
 *   boxes_host  out: (n,4) int32 [x1,y1,x2,y2] = the padded boxes the decode scales by (may be NULL)
 *   boxes_dev   out: the same on the device, ready for hrn_forward(..., HRN_BOX_I32, ...) (may be NULL)
 *   variant     HRN_CROP_PAD (0): single-image path, aspect ratio corrected by zero padding (:243-276);
 *               HRN_CROP_CLAMP (1): the batch path's enlarge-and-clamp, no padding (SimpleHRNet.py:383-412) -- call once
 *               per image of the stack
 * Boxes must lie inside the frame after rounding (the detector wrappers clamp them: YOLOv3.py:49-56) and be
 * non-degenerate; otherwise the call fails (the reference would wrap around / divide by zero). */
enum { HRN_CROP_PAD = 0, HRN_CROP_CLAMP = 1 };
int hrn_preprocess_frame(hrn_handle h, const uint8_t *frame_dev, int frame_h, int frame_w, const float *dets_host,
                         int det_stride, int n, int variant, float *images_dev, int32_t *boxes_host, int32_t *boxes_dev,
                         void *stream);

/* Single-person pre-path on the GPU: replaces, for every frame of a call with multiperson=False,
 *   cv2.resize(image, (W, H), interpolation=self.interpolation); cv2.cvtColor(image, cv2.COLOR_BGR2RGB); self.transform(image)
 * (SimpleHRNet.py:213-222 for one frame, :355-366 for a stack; default interpolation cv2.INTER_CUBIC, :27) and writes the
 * (n,3,H,W) fp32 batch hrn_forward reads.  The boxes of this path are the whole frame: [0, 0, frame_w, frame_h] (:223, :369).
 *   frames_dev     (n, frame_h, frame_w, 3) uint8 BGR, device
 *   interpolation  HRN_INTER_* = the cv2.INTER_* value of the same name; anything else fails (the reference would pass it on)
 * OpenCV is not available where this library is built and tested: the arithmetic follows the published generic 8-bit path of
 * modules/imgproc/src/resize.cpp (oracle/cv2_resize_oracle.py restates it, the kernel equals that restatement bit for bit);
 * equality with a given cv2 build -- IPP / OpenCL builds differ among themselves -- is NOT pinned.  Specifically, for
 * HRN_INTER_CUBIC this is OpenCV's SCALAR cubic path (float32 coefficients, saturate_cast<short>(c * 2048), one rounding after
 * the vertical pass); the SIMD (AVX2 / NEON) builds shipped in the opencv-python wheels round the vertical pass differently and
 * can differ from it by +-1 grey level per sample (nearest and linear are bit-equal).  A deployment that needs equality with ITS
 * cv2 checks it there: tests/golden/make_cv2_golden.py produces the fixture wherever opencv-python is installed. */
enum { HRN_INTER_NEAREST = 0, HRN_INTER_LINEAR = 1, HRN_INTER_CUBIC = 2 };
int hrn_resize_frames(hrn_handle h, const uint8_t *frames_dev, int n, int frame_h, int frame_w, int interpolation,
                      float *images_dev, void *stream);

/* Flip test-time augmentation + evaluation decode (testing/Test.py:132-140,
 * training/COCO.py:206-230, misc/utils.py:9-29 flip_tensor / flip_back, :125-151 get_max_preds, :154-175 the
 * post-processing of get_final_preds):
 *   heatmaps = (model(images) + flip_back(model(flip(images)), flip_pairs)) * 0.5      -> heatmaps_dev (n,J,h,w)
 *   preds    = arg-max of each map as (x, y) in heat-map pixels, zero where the maximum is <= 0, moved a quarter
 *              pixel towards the higher neighbour when post_processing != 0            -> preds_dev (n,J,2)
 *   maxvals  = the maxima                                                              -> maxvals_dev (n,J)
 * flip_pairs_host: npairs x 2 joint indices that swap under mirroring (COCO: datasets/COCO.py:113).  The inverse
 * affine of get_final_preds (transform_preds, cv2) stays with the caller. */
int hrn_forward_flip_tta(hrn_handle h, const void *images_dev, int n, const int32_t *flip_pairs_host, int npairs,
                         int post_processing, float *heatmaps_dev, float *preds_dev, float *maxvals_dev, void *stream);

/* Greedy IoU non-maximum suppression -- the reference's only native component:
 * `void _nms(int *keep_out, int *num_out, const float *boxes_host, int boxes_num, int boxes_dim, float thresh,
 * int device_id)` (misc/nms/gpu_nms.hpp; kernel misc/nms/nms_kernel.cu:33-77, caller misc/nms/gpu_nms.pyx:19-34).
 * Same contract: boxes sorted by score descending, rows [x1,y1,x2,y2,score,...], IoU with the +1 pixel convention,
 * a box is dropped when its IoU with an earlier kept box is > thresh; keep_out receives the kept row indices in
 * order, *num_out their count.  Needs no handle.  Returns 0 or an error code (hrn_nms_last_error()). */
/* SYNCHRONOUS like the reference's _nms (host boxes in, host indices out: two blocking copies on the default stream around
 * the kernels) -- it is the one entry point of this ABI that is not stream-ordered.  At most 65536 boxes (the n x n/64
 * suppression mask is the scratch that grows quadratically: 512 MiB there); the caller's current device is restored.
 * Scratch is kept per device between calls (the reference allocates and frees per call); hrn_nms_release(device_id) frees
 * it (device_id < 0: on every device). */
int hrn_nms(int32_t *keep_out, int32_t *num_out, const float *boxes_host, int boxes_num, int boxes_dim,
            float nms_overlap_thresh, int device_id);
int hrn_nms_release(int device_id);
const char *hrn_nms_last_error(void);

/* ---- pose post-processing on the host (O(people^2 * joints) on a handful of skeletons: host code in the reference,
 * host code here; numpy's float64 / float32 arithmetic and summation order reproduced) ------------------------------
 * OKS non-maximum suppression, misc/nms/nms.py:97-122 (`oks_nms`) and :138-180 (`soft_oks_nms`; at most 20 kept, gaussian
 * rescoring), called per image by datasets/COCO.py:371-374.  kpts: n x J x (x, y, score) float64 (`keypoints.flatten()`),
 * order: the indices by descending score (`scores.argsort()[::-1]`, done by the caller with numpy as the reference does),
 * sigmas: J values or NULL for COCO's 17, in_vis_thre: NaN for None.  keep_out (n entries) receives the kept indices. */
int hrn_oks_nms(int32_t *keep_out, int32_t *num_out, const double *kpts, const double *areas, const int32_t *order, int n, int J,
                double thresh, const double *sigmas, double in_vis_thre);
int hrn_soft_oks_nms(int32_t *keep_out, int32_t *num_out, const double *kpts, const double *areas, const double *scores_sorted,
                     const int32_t *order, int n, int J, double thresh, const double *sigmas, double in_vis_thre);
/* Tracker, misc/utils.py:372-384 (`compute_similarity_matrices`: box IoU :318-334 and OKS :341-369 of every current
 * skeleton against every previous one).  boxes: n x (x1, y1, x2, y2) as float64 (exact for the int32 boxes of the
 * multi-person path), poses: n x J x (y, x, confidence) float32 as predict() returns them; outputs na x nb float32. */
int hrn_pose_similarity(const double *boxes_a, const float *poses_a, int na, const double *boxes_b, const float *poses_b, int nb, int J,
                        float *sim_bbox, float *sim_pose);
/* The optimal assignment `munkres.Munkres().compute(cost)` returns (misc/utils.py:406-407): minimum total cost, every
 * row matched when rows <= cols, otherwise every column; row_to_col[r] = column or -1. */
int hrn_assignment(const double *cost, int rows, int cols, int32_t *row_to_col);

/* Debug tap -- test infrastructure hook for the per-stage parity tests (tests/test_bf16_pin.py), never needed in production.
 * The reference's counterpart is a forward hook on a sub-module (`module.register_forward_hook`, torch.nn.Module): the value
 * of an intermediate of HRNet.forward (models_/hrnet.py:157-189).  A tap is a tensor some launch of the pass writes to HBM:
 *   "stem"                       conv1 + bn1 + ReLU (hrnet.py:158-160)
 *   "<state_dict prefix>"        every convolution, e.g. "layer1.0.conv3", "stage3.1.branches.2.0.conv2",
 *                                "stage4.0.fuse_layers.3.0.2.0", "transition2.2.0.0": its output after the folded BatchNorm,
 *                                the residual and the ReLU, as stored (bf16 engine: the stored bf16 values, widened exactly)
 *   "<stage>.fuse.<i>"           the i-th output of a StageModule (hrnet.py:60-69), e.g. "stage3.2.fuse.0"
 * hrn_forward_tap runs ONE micro-batch (1 <= n <= max_batch) and copies crops crop0, crop0 + crop_step, ... (ncrops of them)
 * of the named tensor to dst_dev as (ncrops, C, H, W) fp32 right after the launch that completes it; heatmaps_dev
 * (n,J,h,w) may be NULL.  A tensor
 * the plan keeps on-chip at this batch size (conv1 of a fused BasicBlock; the projection shortcut of layer1.0) has no tap:
 * the call fails and says so. */
typedef struct {
    char name[96];
    int32_t c, h, w;
    int32_t conv_index;   /* index for hrn_get_conv_info, or -1 (stem / fuse outputs) */
} hrn_tap_info;
int hrn_tap_count(hrn_handle h);
int hrn_get_tap_info(hrn_handle h, int index, hrn_tap_info *out);
int hrn_forward_tap(hrn_handle h, const void *images_dev, int n, const char *tap_name, int crop0, int ncrops, int crop_step,
                    float *dst_dev, float *heatmaps_dev, void *stream);

/* Introspection used by tests, bench.py and the roofline accounting. */
int hrn_conv_count(hrn_handle h);
int hrn_get_conv_info(hrn_handle h, int index, hrn_conv_info *out);
double hrn_flops_per_crop(hrn_handle h);            /* 2*MAC, convolutions only              */
int64_t hrn_workspace_bytes(hrn_handle h);
/* Block maps / descriptor arrays built and uploaded since the handle was created.  They depend on the micro-batch size
 * only; the handle keeps the four most recent sizes, so a call pattern that alternates a few sizes (every predict() whose
 * n is not a multiple of max_batch: SimpleHRNet.py:285-294 runs a short last chunk) stops rebuilding after its first
 * pass over each size -- the test of that property reads this counter. */
int64_t hrn_map_rebuilds(hrn_handle h);
/* The number of entries of the static launch list (grouped launches count once).  A small call may issue more kernels: a
 * stride-2 slab group with too few tiles runs its convolutions on the generic kernel (one or more launches), a debug tap adds one. */
int hrn_launches_per_pass(hrn_handle h);
/* 1 when the handle runs the stem (conv1 + bn1 + ReLU + conv2 + bn2 + ReLU, models_/hrnet.py:158-163) as ONE kernel that
 * keeps conv1's output in LDS (bf16 HRNet handles whose crop width fits; bit-identical to the two launches, which remain
 * the path of hrn_forward_tap("stem") and of HRN_DISABLE_STEM_FUSE=1); hrn_launches_per_pass counts it as one launch. */
int hrn_stem_fused(hrn_handle h);
/* 1 when convolution `index` (hrn_get_conv_info numbering; algo 3 = the 96-cout form of the BasicBlock kernel) enumerates REAL
 * pixels only: its M tiles are runs of h * w * n pixels in (image, row, column) order instead of runs of flat rows of the padded
 * layout, so no matrix instruction is spent on the pad column / pad row (9 % of the 24x18 grid, 17 % of 12x9).  Same input layout,
 * same arithmetic per output pixel: bit-identical to the flat enumeration (HRN_DISABLE_COMPACT=1). */
int hrn_conv_compact(hrn_handle h, int index);
/* The HRN_* environment switches (same-box A/B runs, bit-identity tests) this handle saw when it was
 * created, as "NAME=value;..." -- always "" in production: the library ignores every HRN_* variable unless the process opts in
 * with HRN_DEBUG_ENV=1 (the test suite, tools/ab.sh).  Then they are read at hrn_create only, never during a call. */
const char *hrn_switches(hrn_handle h);
/* Debug: the number of elements at pad / guard positions of the activation workspace that are not zero (synchronises the
 * device; -1 on error or on a plan-only handle).  The layout's invariant -- every 3x3 convolution's zero padding is the pad
 * column / pad row shared by neighbouring rows and images, written as zeros or never touched by every kernel -- says 0 after
 * any sequence of calls; the GPU tests check it after every path of the engine has run. */
int64_t hrn_debug_pad_violations(hrn_handle h);
/* Block map of the `group`-th grouped BasicBlock launch for a call of n crops, as the host would upload it (works on
 * plan-only handles: the CPU tests check that every (conv, cout tile, M tile) is covered exactly once).  Per block six
 * int32: descriptor, cout tile, M tiles walked, first M tile, pixels per M tile, flags (1 = fused BasicBlock, 2 =
 * small tiles).  members[d] = convolution index of descriptor d (+ 2^30: the fused form, which also computes
 * the block's conv2).  Returns the number of blocks (may exceed capacity), -1 for a bad group / n. */
int hrn_plan_block_map(hrn_handle h, int group, int n, int reverse, int32_t *blocks, int capacity, int32_t *members,
                       int member_capacity);
/* Likewise for the `group`-th grouped launch of the generic kernel: per block three int32 (descriptor, cout tile, M tile;
 * M tiles past the end are alignment padding and return at once), members[d] = convolution index, *pixels_per_tile =
 * 64 * (fragments per wave chosen for n). */
int hrn_plan_direct_map(hrn_handle h, int group, int n, int32_t *blocks, int capacity, int32_t *members, int member_capacity,
                        int32_t *pixels_per_tile);
/* Likewise for the `group`-th launch of the stride-2 slab kernel (conv_s2.hip): per block three int32 (problem, tiles walked,
 * first tile; a tile = `rows` output rows of one image, numbered image-major); per part five int32 (problem, convolution
 * index, 48-cout tile of it, output rows per tile, tiles per image); *active = 1 when a call of n crops takes this kernel
 * (0: too few tiles, the same convolutions run on the generic kernel).  Returns blocks | (parts << 20), -1 for a bad group / n.
 * group = -1: the fused stem kernel's map (hrn_stem_fused; stem_fused.hip): conv2 with ONE output row per tile, two parts of 32
 * couts, at most one block per CU, every block an equal run of tiles. */
int hrn_plan_s2_map(hrn_handle h, int group, int n, int32_t *blocks, int capacity, int32_t *parts, int part_capacity,
                    int32_t *active);
/* per-kernel HIP-event timing of one pass (dominant-kernel roofline in bench.py):
 * runs one micro-batch of n crops and returns, for conv i, its device time in ms. */
int hrn_profile_pass(hrn_handle h, const void *images_dev, int n, float *conv_ms, int conv_ms_len,
                     float *other_ms /* [4]: stem, fuse, head, decode */, void *stream);
const char *hrn_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HRNET_MI355_H */
