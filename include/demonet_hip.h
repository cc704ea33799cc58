/* demonet_hip.h -- C ABI of the MI355X (gfx950) SSD inference library.
 *
 * The reference (zhiqwang/demonet) has no FFI/operator layer: its seam is the Python factory ->
 * nn.Module.forward contract (demonet/models/generalized_ssd.py:271-349). This header is the native
 * boundary a maintainer binds instead of calling `SSD.forward` (see INTEGRATION.md for the ctypes stub):
 * plain pointers and sizes only, no torch types. All `*_dev` pointers are device (HBM) pointers owned by
 * the caller; the library owns the plan and its weight arena. No entry point below allocates device
 * memory or synchronises the device except dn_create / dn_destroy.
 *
 * Every function returns 0 on success, a negative DN_E_* code on failure; dn_last_error() gives the text.
 */
#ifndef DEMONET_HIP_H
#define DEMONET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DN_ABI_VERSION 1
#define DN_API __attribute__((visibility("default")))

enum { DN_OK = 0, DN_E_INVALID = -1, DN_E_HIP = -2, DN_E_WORKSPACE = -3, DN_E_UNSUPPORTED = -4 };

/* activation fused behind a convolution (reference: mobilenetv3.py:72, ssd_mobilenetv3.py:31,41) */
enum { DN_ACT_NONE = 0, DN_ACT_RELU = 1, DN_ACT_RELU6 = 2, DN_ACT_HSWISH = 3 };

/* op kinds of the lowered graph (demonet_amd/spec.py IR) */
enum { DN_OP_STEM = 1, DN_OP_PW = 2, DN_OP_DW = 3, DN_OP_SE = 4, DN_OP_CONV = 5, DN_OP_MAXPOOL = 6, DN_OP_L2NORM = 7 };

/* tensor kinds: NHWC fp16 activation | NCHW fp32 image | per-image fp32 vector [c] | fp32 pooled partial sums [blocks][c] */
enum { DN_T_ACT = 0, DN_T_IMAGE = 1, DN_T_VEC = 2, DN_T_POOL = 3 };

typedef struct dn_tensor_desc {
    int32_t c, h, w, kind;
} dn_tensor_desc;

typedef struct dn_op_desc {
    int32_t type;                       /* DN_OP_* */
    int32_t in, out;                    /* tensor ids */
    int32_t residual;                   /* PW: tensor added after BN (mobilenetv3.py:97-98); -1 none */
    int32_t se;                         /* PW: DN_T_VEC tensor scaling the input channels per image; -1 none */
    int32_t pool;                       /* DW: DN_T_POOL tensor receiving per-(image,channel) sums; -1 none */
    int32_t cin, cout, k, stride, pad, dil, act;
    int32_t head;                       /* 0 none, 1 class logits, 2 box regression: write fp32 into the head arrays */
    int32_t level;                      /* pyramid level of a head op */
    int32_t squeeze;                    /* SE: squeeze width; pooled pixel count is taken from the producer */
    int32_t ceil_mode;                  /* MAXPOOL */
    int32_t pool_pixels;                /* SE: H*W of the pooled map (mean divisor) */
    int32_t reserved[2];
    int64_t w_off, b_off, w2_off, b2_off;   /* byte offsets into the weight blob; -1 none.
                                               PW/CONV: w = fp16 [cout][k*k*cin] (tap-major, channel-minor), b = fp32 [cout]
                                               PW (optional): w2 = the same weights in MFMA-fragment order
                                                 [ceil(cout/32)][ceil(cin/16)][2 k-halves][32 channels][8] fp16, zero beyond cout / cin; -1: none
                                               DW/STEM: w = fp16 [k*k][c] / fp32 [k*k*3][cout], b = fp32 [c]
                                               SE: w = fp16 fc1 weight TRANSPOSED [c][squeeze], b = fp32 fc1 bias, w2 = fp16 fc2 weight TRANSPOSED [squeeze][c], b2 = fp32 fc2 bias (c, squeeze even)
                                               L2NORM: w = fp32 scale [c] */
} dn_op_desc;

typedef struct dn_model_desc {
    int32_t abi_version;                /* DN_ABI_VERSION */
    int32_t n_tensors, n_ops;
    const dn_tensor_desc* tensors;
    const dn_op_desc* ops;
    int32_t input_tensor;
    int32_t image_h, image_w;           /* fixed network input size (generalized_ssd.py:190-191) */
    float mean[3], std[3];              /* transform.py:129-138 */
    int32_t num_classes;                /* including background class 0 */
    int32_t n_levels;
    int32_t level_tensor[8];            /* feature-map tensor id per pyramid level */
    int32_t anchors_per_loc[8];
    int32_t num_anchors;
    const float* anchors;               /* host, [num_anchors][4] xyxy pixels (anchor_utils.py:111-126) */
    float score_thresh, nms_thresh;     /* generalized_ssd.py:158-162 */
    int32_t detections_per_img, topk_candidates;
} dn_model_desc;

typedef struct dn_plan dn_plan;

/* Build a plan: validates the graph, uploads `weights` (host blob addressed by the op offsets) and anchors. */
DN_API int dn_create(const dn_model_desc* desc, const void* weights, size_t weight_bytes, dn_plan** out);
DN_API void dn_destroy(dn_plan* plan);

/* Bytes of caller-provided device scratch needed for a batch of n images. */
DN_API size_t dn_workspace_bytes(const dn_plan* plan, int n);

/* The whole hot path, replacing SSD.forward (eval):  images_dev = [n][3][h][w] fp32 in [0,1], NCHW, contiguous
 * (the stacked form of the reference's List[Tensor[3,H,W]]); (h, w) may differ from the network size, in which
 * case the bilinear resize of transform.py:27-53 runs first and boxes are mapped back (transform.py:278-292).
 * Outputs (device): boxes [n][D][4] fp32 xyxy, scores [n][D] fp32, labels [n][D] int64, counts [n] int32, with
 * D = detections_per_img; rows >= counts[i] are zero. Asynchronous on `stream` (a hipStream_t). */
DN_API int dn_forward(dn_plan* plan, const float* images_dev, int n, int h, int w,
               float* boxes_dev, float* scores_dev, int64_t* labels_dev, int32_t* counts_dev,
               void* workspace_dev, size_t workspace_bytes, void* stream);

/* The same with the decoder's output as input -- the step just before the path: images_dev = [n][h][w][3] uint8, HWC,
 * RGB. x/255 (ToTensor), the bilinear resize of transform.py:27-53 and the HWC -> planar conversion run as one pass ahead of the
 * stem (which normalises on load); results are identical to dn_forward on float(images)/255 in NCHW.
 * The call marks the plan's current input as uint8 for its duration (shared plan state): like dn_forward, one plan serves one host
 * thread at a time. */
DN_API int dn_forward_u8(dn_plan* plan, const uint8_t* images_dev, int n, int h, int w,
                  float* boxes_dev, float* scores_dev, int64_t* labels_dev, int32_t* counts_dev,
                  void* workspace_dev, size_t workspace_bytes, void* stream);

/* Backbone + heads only (no post-process). The head outputs live inside the workspace; query them with
 * dn_head_outputs:  cls_logits [n][A][K] fp32, bbox_regression [n][A][4] fp32 (generalized_ssd.py:60-74). Valid after
 * dn_forward_heads only: dn_forward may compute softmax / box decode inside the head launch and then never writes the logits
 * of the large pyramid levels -- dn_head_outputs returns DN_E_INVALID for a workspace whose last forward was dn_forward. */
DN_API int dn_forward_heads(dn_plan* plan, const float* images_dev, int n, int h, int w,
                     void* workspace_dev, size_t workspace_bytes, void* stream);
DN_API int dn_head_outputs(const dn_plan* plan, void* workspace_dev, int n, float** cls_logits_dev, float** bbox_regression_dev);
/* Device address/size of an intermediate tensor inside the workspace (parity tests on feature maps). */
DN_API int dn_tensor_ptr(const dn_plan* plan, void* workspace_dev, int n, int tensor_id, void** ptr, size_t* bytes);

/* Post-process only, replacing SSD.postprocess_detections (generalized_ssd.py:351-397) + transform.postprocess:
 * softmax -> decode (BoxCoder weights 10,10,5,5) -> clip -> per-class score>thr & top-k -> hard NMS (IoU > thr)
 * -> global top-D by score.  kept_anchor_dev (optional, may be NULL): [n][D] int32 anchor index per detection. */
DN_API size_t dn_postprocess_workspace_bytes(int n, int num_anchors, int num_classes, int topk_candidates, int detections_per_img);
DN_API int dn_postprocess(const float* cls_logits_dev, const float* bbox_regression_dev, const float* anchors_dev,
                   int n, int num_anchors, int num_classes,
                   float image_h, float image_w, const float* scale_xy_dev /* [n][2] (w,h) ratios or NULL */,
                   float score_thresh, float nms_thresh, int topk_candidates, int detections_per_img,
                   float* boxes_dev, float* scores_dev, int64_t* labels_dev, int32_t* counts_dev,
                   int32_t* kept_anchor_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* Single-kernel entry points (unit parity tests, micro-benchmarks, roofline measurement).
 * x: [m][cin] fp16 (NHWC rows), w: [cout][cin] fp16, bias fp32 [cout], residual [m][cout] fp16 or NULL,
 * w_frag (optional): the same weights in MFMA-fragment order (dn_op_desc PW w2) -- enables the strip kernel,
 * se_scale fp32 [m/hw][cin] or NULL, out fp16 [m][cout] (out_fp32 != 0: fp32, addressed
 * (row/hw)*out_img_stride + (row%hw)*cout + c). */
DN_API int dn_pointwise_conv(const void* x_dev, const void* w_dev, const void* w_frag_dev, const float* bias_dev,
                      const void* residual_dev, const float* se_scale_dev, void* out_dev, int m, int cin, int cout, int hw,
                      int act, int out_fp32, int64_t out_img_stride, void* stream);
/* Dense kxk convolution + bias + activation (the VGG path: vgg16 "D" convs, fc6 dilated 3x3, fc7 1x1, extras, ssd_vgg16.py:30-109).
 * x: [n][h][w][cin] fp16, w: [cout][k][k][cin] fp16, bias fp32 [cout], out [n][ho][wo][cout] fp16; cin % 32 == 0, cout % 4 == 0.
 * zeros (optional): >= 16 zero bytes on the device; with it, layers with cin % 64 == 0 and cout % 256 == 0 run on the
 * 256x256-tile kernel (taps outside the image are read from there). */
DN_API int dn_dense_conv(const void* x_dev, const void* w_dev, const float* bias_dev, const void* zeros_dev, void* out_dev,
                  int n, int h, int w, int cin, int cout, int k, int stride, int pad, int dil, int act, void* stream);
/* x: [n][h][w][c] fp16, w: [k*k][c] fp16, bias fp32 [c], out [n][ho][wo][c] fp16 */
DN_API int dn_depthwise_conv(const void* x_dev, const void* w_dev, const float* bias_dev, void* out_dev,
                      int n, int h, int w, int c, int k, int stride, int pad, int act, void* stream);

/* Fused inverted-residual stages (expdw.hip): [1x1 expand w1/b1 + act1] -> depthwise kxk wd/bd + act2 -> [1x1 project w3/b3
 * (+ residual = x)], BN folded, padding (k-1)/2; replaces the ConvBNActivation chain of InvertedResidual.forward
 * (mobilenetv3.py:72-99) / _extra_block (ssd_mobilenetv3.py:39-54). The expanded / depthwise activations stay on chip.
 * w1 may be NULL (no expand: cexp == cin), w3 may be NULL (stop after the depthwise stage: out is [n][ho][wo][cexp] and
 * pool_partial (optional) receives [n][tiles][cexp] fp32 channel sums, tiles = dn_expand_depthwise_tiles(ho, wo, stride));
 * not both. x: [n][h][w][cin] fp16, w1 [cexp][cin], wd [k*k][cexp], w3 [cout][cexp] fp16. cin <= 128, k in {3,5}, stride in
 * {1,2}; with w3: (tile pixels / 32) * (cout / 32) <= 8 and no pooling. */
DN_API int dn_expand_depthwise(const void* x_dev, const void* w1_dev, const float* b1_dev, const void* wd_dev, const float* bd_dev,
                        const void* w3_dev, const float* b3_dev, void* out_dev, float* pool_partial_dev,
                        int n, int h, int w, int cin, int cexp, int cout, int k, int stride, int act1, int act2, int has_res,
                        void* stream);
DN_API int dn_expand_depthwise_tiles(int ho, int wo, int stride);

/* 1 = replay the launch sequence from a cached hipGraph keyed by (n, pointers) [default], 0 = eager launches */
DN_API int dn_set_graph_mode(dn_plan* plan, int enabled);

/* Timing hook for bench.py: average device time in ms of the op at `op_index` over the forwards recorded since
 * dn_profile_begin (HIP events on the forward stream, eager mode). */
DN_API int dn_profile_begin(dn_plan* plan);
DN_API int dn_profile_end(dn_plan* plan, float* ms_per_op /* [n_ops + 4] : ops..., softmax/decode (0 when the fused head launch does it), cut-off + select/NMS, merge, fallback select + merge */, int capacity);
/* After a profiled forward: the label of the kernel launch op `op_index` took part in (same spelling as rocprofv3's kernel
 * names, e.g. "pw_kernel<128,64,4,1,false,32>") and the op whose ms_per_op slot holds that launch's time (grouped launches
 * serve several ops; their members report the same owner). */
DN_API int dn_profile_op_info(const dn_plan* plan, int op_index, char* kernel, int capacity, int32_t* owner);

/* Optional extra output of dn_forward for the multi-GPU gather: when non-NULL, the final merge kernel also writes
 * packed_dev [n][D+1][6] fp32 -- rows (x1,y1,x2,y2,score,label), row D = (count,0,0,0,0,0) -- i.e. the fixed-shape payload of
 * the detections all-gather (the analogue of util/misc.py:75-115 all_gather of pickled results). NULL disables it. */
DN_API int dn_set_packed_output(dn_plan* plan, float* packed_dev);

/* Number of independent sub-batch launch chains a forward of n images is issued as (parallel hipGraph branches; 1 = a single
 * chain). Every kernel then runs once per sub-batch on ~n/split images; results are identical to the unsplit forward. */
DN_API int dn_batch_split(const dn_plan* plan, int n);
/* Overrides that choice: chains > 0 = every forward is issued as min(chains, n) sub-batch chains, 0 = back to the automatic
 * choice. For callers that keep several forwards in flight on streams of their own (one workspace and output set per forward):
 * whole-batch chains of different forwards overlap better than the half-size chains of one. Changes dn_workspace_bytes and drops
 * the cached graphs: call it before sizing workspaces, with no forward of this plan in flight. */
DN_API int dn_set_chains(dn_plan* plan, int chains);

/* SSD training loss, forward value only (no gradients). Replaces, per batch: the matching of
 * generalized_ssd.py:316-330 (torchvision box_iou -> SSDMatcher, _utils.py:264-294,348-362) and SSD.compute_loss
 * (generalized_ssd.py:210-269: encode_boxes _utils.py:100-133, smooth_l1_loss(sum), cross_entropy(none), hard negative mining with
 * neg_to_pos_ratio * (#label > 0) negatives per image, both sums / max(1, #matched anchors)).
 * cls_logits [n][A][K] fp32, bbox_regression [n][A][4] fp32, anchors [A][4] fp32 xyxy (the same for every image, as
 * DefaultBoxGenerator produces them), gt_boxes [n][gmax][4] fp32 xyxy / gt_labels [n][gmax] int64 padded, gt_counts [n] int32
 * (0 = an image without boxes: all background). matched_idxs_dev (optional, [n][A] int64) receives the matched gt index or -1;
 * losses_dev [2] = {bbox_regression, classification}. All pointers are device pointers; asynchronous on `stream`. */
DN_API size_t dn_ssd_loss_workspace_bytes(int n, int num_anchors);
DN_API int dn_ssd_loss(const float* cls_logits_dev, const float* bbox_regression_dev, const float* anchors_dev,
                       const float* gt_boxes_dev, const int64_t* gt_labels_dev, const int32_t* gt_counts_dev,
                       int n, int num_anchors, int num_classes, int gmax, float iou_thresh, float neg_to_pos_ratio,
                       int64_t* matched_idxs_dev, float* losses_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

DN_API const char* dn_last_error(void);
DN_API int dn_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DEMONET_HIP_H */
