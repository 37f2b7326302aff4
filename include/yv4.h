/*
 * yv4.h -- C-ABI of libyv4_hip.so: the MI355X (gfx950) YOLOv4 hot path.
 *
 * This is the drop-in boundary.  Every entry point is `extern "C"`, takes plain
 * device pointers + sizes + a hipStream_t passed as void*, allocates nothing,
 * never synchronises the host, and returns 0 on success or a negative YV4_E_*
 * code (yv4_last_error() holds the message of the last failure on the calling
 * thread).  No torch types cross this boundary.
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   yv4_mish_fwd / yv4_mish_bwd   mmdet/ops/mish_cuda/src/mish.cc:14-39
 *                                 (pybind mish_forward / mish_backward),
 *                                 math mmdet/ops/mish_cuda/src/mish.h:16-29,
 *                                 kernels src/kernel/mish_cuda.cu:25-71
 *   yv4_conv_bn_act_fwd           mmcv ConvModule as used by
 *                                 mmdet/models/backbones/darknetcsp.py:15-35
 *                                 (Conv2d -> BN(eval) -> act), residual add of
 *                                 darknetcsp.py:60-64, CSP-level cat->BN->act of
 *                                 darknetcsp.py:106-109,149-153,220-229 and the
 *                                 biased head conv yolocsp_head.py:180-185
 *   yv4_spp_pool_fwd              darknetcsp.py:176-181,203-206,222-226
 *                                 (MaxPool2d k=5,9,13 s=1 + cat)
 *   yv4_resample_nearest_fwd      necks/yolo_neck_csp.py:213-219,229
 *                                 (F.interpolate nearest + torch.cat)
 *   yv4_decode_filter             dense_heads/yolocsp_head.py:255-294,357-372,
 *                                 core/bbox/coder/yolov4_bbox_coder.py:39-67,
 *                                 core/anchor/anchor_generator.py:207-270,
 *                                 core/post_processing/bbox_nms.py:36-67
 *   yv4_nms_images / yv4_nms_prepare
 *                                 mmcv.ops.nms.batched_nms called at
 *                                 core/post_processing/bbox_nms.py:84-88
 *   yv4_nchw_to_nhwc / yv4_nhwc_to_nchw
 *                                 layout adaptors at the module boundary (the
 *                                 reference is NCHW, yolocsp_head.py:264 permutes)
 */
#ifndef YV4_H_
#define YV4_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YV4_ABI_VERSION 7

/* error codes */
#define YV4_OK 0
#define YV4_E_INVALID (-1)   /* bad argument (null pointer, bad shape, bad enum) */
#define YV4_E_UNSUPPORTED (-2) /* valid request this build has no kernel for */
#define YV4_E_LAUNCH (-3)    /* hipLaunch / hip runtime error */
#define YV4_E_CAPACITY (-4)  /* caller-provided workspace too small */

/* element types of the standalone activation op */
#define YV4_F32 0
#define YV4_F16 1
#define YV4_BF16 2
#define YV4_F64 3

/* activations of the fused conv epilogue (Mish everywhere in v4/v5,
 * LeakyReLU(slope) in the v3 path, Swish when a config overrides act_cfg) */
#define YV4_ACT_NONE 0
#define YV4_ACT_MISH 1
#define YV4_ACT_LEAKY 2
#define YV4_ACT_SWISH 3

int yv4_abi_version(void);
const char* yv4_last_error(void);
/* Name of the gfx target the kernels were compiled for ("gfx950"). */
const char* yv4_arch(void);

/* ---- Mish (standalone op; on the fast path Mish is a conv epilogue) -------
 * out[i] = x*tanh(x < 20 ? log1p(exp(x)) : x);   n elements, contiguous.
 * bwd: gin = gout * (x*(1-tsp^2)*(1-exp(-sp)) + tsp), from the saved INPUT. */
int yv4_mish_fwd(const void* in, void* out, size_t n, int dtype, void* stream);
int yv4_mish_bwd(const void* gout, const void* in, void* gin, size_t n,
                 int dtype, void* stream);
/* The same op on HOST memory (ABI 6): mish.cc:14-33 dispatches on input.is_cuda() and a CPU tensor runs
 * mish_cpu_kernel's loop (src/kernel/mish_cpu.cc:6-29) over mish.h:16-29 -- float and double only (Half / BFloat16 are
 * not dispatched there: YV4_E_UNSUPPORTED), the float forward evaluated in double as that build does.  Plain
 * single-threaded loops over libm.  Not on any plan's or train step's path -- the op's CPU behaviour, nothing else. */
int yv4_mish_fwd_host(const void* in, void* out, size_t n, int dtype);
int yv4_mish_bwd_host(const void* gout, const void* in, void* gin, size_t n, int dtype);

/* ---- layout adaptors -------------------------------------------------------
 * src NCHW (N,C,H,W) contiguous fp32 -> dst NHWC with `dst_cstride` channels per
 * pixel, written at channel offset dst_coff; channels [C, C+zero_pad) are zeroed
 * (the stem pads Cin 3 -> 4). */
int yv4_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W,
                     int dst_cstride, int dst_coff, int zero_pad, void* stream);
int yv4_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W,
                     int src_cstride, int src_coff, void* stream);

/* ---- fused conv ------------------------------------------------------------
 * Implicit-GEMM convolution on fp32 MFMA, NHWC activations:
 *   acc[m][co] = sum_{kh,kw,ci} x[n, ho*stride-pad+kh, wo*stride-pad+kw, ci]
 *                               * w[co][(kh*KW+kw)*Cin + ci]
 *   v = acc*scale1[co] + shift1[co];  v = act1(v);
 *   if (residual) v += residual[m][co];
 *   if (scale2)   v = act2(v*scale2[co] + shift2[co]);
 *   y[m*y_cstride + y_coff + co] = v
 * scale1/shift1 carry the folded eval-mode BN (or 1 / bias for a bare or biased
 * conv); scale2/shift2 carry the CSP-level BN applied to the concat half this
 * conv produces.  x / residual / y are channel-strided views so that producers
 * write straight into concat buffers (torch.cat disappears). Cin % 4 == 0,
 * x_cstride % 4 == 0, x_coff % 4 == 0 are required (16-byte loads). */
typedef struct yv4_conv_desc {
  int32_t N, H, W, Cin;      /* input  */
  int32_t Ho, Wo, Cout;      /* output */
  int32_t KH, KW, stride, pad;
  int32_t x_cstride, x_coff; /* floats per input pixel / channel offset  */
  int32_t y_cstride, y_coff;
  int32_t r_cstride, r_coff; /* residual view (ignored when residual==NULL) */
  int32_t act1, act2;        /* YV4_ACT_* */
  float slope1, slope2;      /* LeakyReLU negative slopes */
  int32_t tile;              /* 0 = auto; else a YV4_TILE_* id (benchmarking) */
  int32_t flags;             /* ABI 7: YV4_CONV_* bits, 0 = none (appended: an ABI-6 caller's struct is a prefix) */
} yv4_conv_desc;

/* flags: the output is written with NON-TEMPORAL stores (the lines are not kept in the XCD's L2).  A per-plan choice:
 * inference plans of the 16-bit wide-tile kernels gain 0.6-1.3 % with it (their outputs are next read by a kernel that
 * streams them once), training loses 0.2-0.7 % (the BatchNorm pass right behind a conv re-reads its output from the
 * cache), the fp32 wide kernels lose 2.8 % (DESIGN, non-temporal stores).  Same bits either way; kernels without a
 * non-temporal form ignore the flag. */
#define YV4_CONV_NT_OUT 1

#define YV4_TILE_AUTO 0
#define YV4_TILE_128x128 1
#define YV4_TILE_128x64 2
#define YV4_TILE_64x64 3
#define YV4_TILE_64x128 4
/* LDS-DMA kernels (need Cin % 32 == 0 and tensors below 4 GiB) */
#define YV4_TILE_DMA_64x64 5
#define YV4_TILE_DMA_128x64 6
#define YV4_TILE_DMA_128x128 7
/* register-only kernel for the 3x3/s1/p1 stem (Cin padded to 4, Cout <= 64) */
#define YV4_TILE_STEM 8
/* 1x1 / stride 1, Cin 64 / 128 / 256, Cout >= 32, no residual: one persistent 8-wave workgroup per CU, the weight slab
 * resident in LDS, wave-private rings of 32-pixel strips; same summation order as the DMA tiles (bit-identical) */
#define YV4_TILE_WS_1x1 9
/* fp32 3x3 / stride 1 / pad 1 with Cin % 32 == 0 and Cout % 16 == 0 on WIDE wave tiles (16 PT pixels x 64 channels per
 * wave on v_mfma_f32_16x16x4_f32, one accumulator set; the fp32 form of YV4_HTILE_W3x3: csrc/conv3x3_wide_f32.hip).
 * NOT bit-identical to the 32x32x2 tiles (the K sum is grouped differently); YV4_TILE_W3x3_SHAPE(i), i = 0..4 (256 x 256,
 * 192 x 256, 128 x 256, 384 x 128, 256 x 128), pins the workgroup tile shape -- yv4_conv_pick_tile returns that form, so a
 * plan that must reproduce another plan's bits can copy its tile ids. */
#define YV4_TILE_W3x3 10
#define YV4_TILE_W3x3_SHAPE(i) (10 + 16 * ((i) + 1))
/* fp32 GENERAL convolution on the same wide wave tiles (csrc/conv_wide_f32.hip, the fp32 form of YV4_HTILE_WIDE): any
 * kernel / stride / padding with Cin % 32 == 0, Cout % 16 == 0 (64 .. 2048), 4-aligned channel strides / offsets; the
 * automatic choice gives it the stride-2 3x3 layers and the deep 1x1 layers.  Not bit-identical to the 32x32x2 tiles;
 * YV4_TILE_WIDE_SHAPE(i) pins the workgroup tile shape as above and is what yv4_conv_pick_tile reports. */
#define YV4_TILE_WIDE 11
#define YV4_TILE_WIDE_SHAPE(i) (11 + 16 * ((i) + 1))
/* (round 2's fp32 ping-pong 3x3 form, removed in round 3 -- DESIGN 9.12 -- used to own id 10; the id now names the wide
 * 3x3 kernel above and is accepted and chosen automatically) */

int yv4_conv_bn_act_fwd(const yv4_conv_desc* d, const float* x, const float* w,
                        const float* scale1, const float* shift1,
                        const float* scale2, const float* shift2,
                        const float* residual, float* y, void* stream);
/* Algorithmic work of one launch, for the roofline: 2*M*Cout*K flops. */
double yv4_conv_flops(const yv4_conv_desc* d);
/* Tile id the auto heuristic picks for d (so callers can report it). */
int yv4_conv_pick_tile(const yv4_conv_desc* d);

/* ---- SPP -------------------------------------------------------------------
 * buf is an NHWC view with `cstride` channels per pixel.  Reads channels
 * [coff, coff+C) and writes MaxPool2d(k, stride 1, pad k/2) for k = 5, 9, 13 to
 * [coff+C, coff+2C), [coff+2C, coff+3C), [coff+3C, coff+4C) -- i.e. the
 * torch.cat([x, mp5, mp9, mp13], 1) buffer is completed in place. C % 4 == 0. */
int yv4_spp_pool_fwd(float* buf, int N, int H, int W, int C, int cstride,
                     int coff, void* stream);

/* ---- nearest resample into a concat buffer ---------------------------------
 * dst[n, y, x, dst_coff + c] = src[n, sy(y), sx(x), src_coff + c] with
 * sy = min(floor(y * Hs / Hd), Hs-1) (torch 'nearest').  Hs==Hd is a plain
 * channel-slice copy. C % 4 == 0. */
int yv4_resample_nearest_fwd(const float* src, float* dst, int N, int Hs, int Ws,
                             int Hd, int Wd, int C, int src_cstride,
                             int src_coff, int dst_cstride, int dst_coff,
                             void* stream);

/* ---- decode + threshold ----------------------------------------------------
 * One call handles every level of a batch.  Level l's pred map is NHWC
 * (N, H_l, W_l, A*(5+num_classes)) fp32.  For anchor-box index
 * j = level_base_l + (y*W_l + x)*A + a of image n:
 *   s = sigmoid(p);  cx = (2 s0 - 1)*stride + x*stride + stride/2 ... (the
 *   reference's literal op order, see csrc/postproc.hip)
 *   boxes[n][j] = decoded box / scale_factor[n]   (if scale_factor != NULL)
 *   conf[n][j]  = s4
 *   for each class c with s_{5+c}*s4 > score_thr: append the candidate key
 *       (~bits(score) << 32 | (j*num_classes + c)) to keys[n*key_cap + ...],
 *       and fold the box into max_coord[n] (the per-image boxes.max()).
 * num_classes == 0 is the class-agnostic head (yolocsp_head.py:155-178,357-360:
 * 5 attributes per box, one score column = conf, flat index = j).
 * topk_keys (N device values from yv4_conf_topk, or NULL) implements `nms_pre`
 * (yolocsp_head.py:349-355): boxes whose (conf, index) key ranks behind the
 * image's k-th are decoded but produce no candidates.
 * counts[n] / max_coord[n] must be zero / -inf-initialised by
 * yv4_decode_reset().  If more than key_cap candidates pass for an image the
 * count keeps counting (so the caller can see the overflow) but keys beyond the
 * capacity are dropped. */
typedef struct yv4_level_desc {
  const float* pred;  /* device pointer, NHWC */
  int32_t H, W;
  int32_t stride;     /* featmap stride in pixels (8/16/32) */
  /* base anchors (x1,y1,x2,y2) of one grid cell, A <= 8 rows, exactly as
   * YOLOAnchorGenerator.gen_single_level_base_anchors builds them
   * (core/anchor/anchor_generator.py:639-665): centre stride/2, float32 */
  float base_anchors[8][4];
} yv4_level_desc;

int yv4_decode_reset(int32_t* counts, float* max_coord, int N, void* stream);
int yv4_decode_filter(const yv4_level_desc* levels, int num_levels, int N, int A,
                      int num_classes, float score_thr,
                      const float* scale_factor /* (N,4) device or NULL */,
                      float* boxes /* (N, total_anchors, 4) */,
                      float* conf /* (N, total_anchors) or NULL */,
                      float* cls /* (N, total_anchors, num_classes) sigmoid, or NULL */,
                      uint64_t* keys, int64_t key_cap, int32_t* counts,
                      float* max_coord, const uint64_t* topk_keys /* (N) or NULL */,
                      void* stream);

/* nms_pre pre-selection: topk_keys[n] = the k-th smallest key
 * (~order(conf) << 32 | anchor index) of image n, i.e. the admission threshold of
 * `conf_pred.topk(k)` with ties broken towards the lower anchor index.  Requires
 * 0 < k < anchors per image (the reference applies top-k only then).  work:
 * yv4_conf_topk_work(N, anchors per image) bytes of device memory. */
size_t yv4_conf_topk_work(int N, int64_t total_anchors);
int yv4_conf_topk(const yv4_level_desc* levels, int num_levels, int N, int A,
                  int num_classes, int k, void* work, uint64_t* topk_keys,
                  void* stream);

/* ---- YOLOv3 head (mmdet/models/dense_heads/yolo_head.py:210-391) ----------------------
 * Same buffers and key format as yv4_decode_filter, YOLOV3Head semantics:
 *   box: cx = (sigmoid(t0) - 0.5)*stride + anchor_cx, w = exp(t2)*anchor_w
 *        (core/bbox/coder/yolo_bbox_coder.py:61-89);
 *   candidates: boxes inside their LEVEL's top-k by objectness (topk_keys_per_level: N*num_levels
 *   values from yv4_conf_topk_levels, or NULL), with objectness >= conf_thr (<= 0: off), classes with
 *   sigmoid(cls) > score_thr; the key carries cls*objectness (multiclass_nms' score_factors,
 *   core/post_processing/bbox_nms.py:52-62).  NMS itself is yv4_nms_images / yv4_nms_split. */
int yv4_decode_filter_v3(const yv4_level_desc* levels, int num_levels, int N, int A,
                         int num_classes, float score_thr, float conf_thr,
                         const float* scale_factor, float* boxes, float* conf, float* cls,
                         uint64_t* keys, int64_t key_cap, int32_t* counts, float* max_coord,
                         const uint64_t* topk_keys_per_level, void* stream);
/* Per-level top-k thresholds: topk_keys[n*num_levels + l] = k-th key of level l of image n, or the
 * all-admitting key when the level has <= k boxes (core/export/onnx_helper.py:45-78). */
size_t yv4_conf_topk_levels_work(int N, int64_t total_anchors, int num_levels);
int yv4_conf_topk_levels(const yv4_level_desc* levels, int num_levels, int N, int A,
                         int num_classes, int k, void* work, uint64_t* topk_keys, void* stream);

/* ---- batched NMS -----------------------------------------------------------
 * Per image n (one workgroup each): sort its counts[n] candidate keys by
 * (score desc, flat index asc), then greedy NMS in that order on the
 * class-offset boxes  box + label*(max_coord[n] + 1)  (mmcv batched_nms
 * semantics), suppressing j when inter/(area_i + area_j - inter) > iou_thr
 * (fp32, IEEE division).  Stops after max_out survivors (multiclass_nms
 * `max_num`).  A candidate's box is boxes[n][flat / fused_classes] and its
 * label flat % fused_classes; with fused_classes == 0 the box is
 * boxes[n][flat] and the label labels[n*label_stride + flat] (the standalone
 * batched_nms op).
 * Outputs per image: out_dets (max_out,5) = x1,y1,x2,y2,score (un-offset
 * boxes), out_labels (max_out) int32, out_index (max_out) int64 = flat index of
 * each survivor, out_count.  Images with counts[n] >= split_thr (mmcv
 * `split_thr`, 10000) or > key_cap take mmcv's per-class path: they are only
 * flagged here (out_count[n] = -1) and must go through yv4_nms_split. */
int yv4_nms_images(uint64_t* keys /* sorted in place */, int64_t key_cap, const int32_t* counts,
                   const float* max_coord, const float* boxes,
                   int64_t boxes_per_image, const int32_t* labels,
                   int64_t label_stride, int fused_classes, int N,
                   float iou_thr, int max_out, int split_thr, float* out_dets,
                   int32_t* out_labels, int64_t* out_index, int32_t* out_count,
                   void* stream);

/* mmcv's n >= split_thr path for ONE image: per-class NMS, survivors re-sorted
 * by (score desc, flat index asc), first max_out returned.  `keys` holds that
 * image's n candidate keys (any order); work must hold yv4_nms_split_work(n)
 * bytes.  n is a host value: the caller reads counts[] back first (the
 * reference syncs at the same place, bbox_nms.py:66). */
size_t yv4_nms_split_work(int64_t n);
int yv4_nms_split(const uint64_t* keys, int64_t n, float max_coord,
                  const float* boxes, const int32_t* labels, int fused_classes,
                  float iou_thr, int max_out, void* work, float* out_dets,
                  int32_t* out_labels, int64_t* out_index, int32_t* out_count,
                  void* stream);

/* The suppression predicate of every NMS entry point, process-wide (set it once, before launching):
 *   YV4_NMS_IOU_DIV (default)  inter / (Sa + Sb - inter) > iou_thr    -- mmcv-full 1.3.x's CPU kernel (nms_cpu) and
 *                                                                        the definition SURVEY 8c fixes; oracle/nms_ref.c
 *   YV4_NMS_IOU_MUL            inter > iou_thr * (Sa + Sb - inter)    -- mmcv-full 1.3.x's CUDA kernel (devIoU), i.e.
 *                                                                        what the reference executes on a GPU
 * Both in fp32 without contraction; they select differently only on pairs whose IoU rounds across iou_thr
 * (tests/golden/nms_boundary.npz).  mmcv is third party and absent from the reference tree (call site
 * mmdet/core/post_processing/bbox_nms.py:84): parity unpinned against either kernel.
 * Returns YV4_OK / YV4_E_ARG. */
#define YV4_NMS_IOU_DIV 0
#define YV4_NMS_IOU_MUL 1
int yv4_nms_set_iou_form(int form);
int yv4_nms_get_iou_form(void);

/* ---- deterministic mode (ABI 6) ----------------------------------------------------------------
 * The reference's training step is bit-reproducible run to run wherever torch's is (torch.nn.BatchNorm2d,
 * mmdet/models/backbones/darknetcsp.py:15-35, sums its statistics in a fixed order); this library's default sums
 * the BatchNorm statistics, the BatchNorm backward's dbeta / dgamma, the loss sums, the positives' row gradients,
 * the head's bias gradients, the SPP backward scatter and the gradient norm with atomics in arrival order: equal
 * to rounding, not to the bit -- and a 110-layer network under batch statistics turns one ulp into percents of
 * the loss within a hundred steps.  yv4_set_deterministic(1) switches all of those accumulations to 64-bit
 * FIXED-POINT integer words (integer addition is associative: the result does not depend on the arrival order;
 * a non-finite or out-of-range addend reads back as NaN) and the gradient norm to per-workgroup partials added
 * in index order.  With the deterministic weight gradient (yv4_conv_wgrad_det, the default of the Python host)
 * two runs of a training step from the same state give the same bits.  Process-wide; switch it between steps.
 * SCRATCH SIZES below ("work", "sums", "dbias", "gpos", "zero_after") are stated for both modes: the
 * deterministic forms need twice the 64-bit words (hi | lo), so callers allocate the larger size always. */
int yv4_set_deterministic(int on);
int yv4_get_deterministic(void);

/* Build candidate keys for the standalone batched_nms op from plain
 * boxes/scores (n candidates of one image): keys[i] = ~bits(score_i)<<32 | i,
 * counts[0] = n, max_coord[0] = max over all box coordinates. */
int yv4_nms_prepare(const float* boxes, const float* scores, int64_t n,
                    uint64_t* keys, int32_t* counts, float* max_coord,
                    void* stream);

/* ---- training side (fp32) -----------------------------------------------------
 * Replaces the autograd of mmcv ConvModule in training mode (cuDNN backward-filter /
 * backward-data, ATen batch_norm fwd/bwd with batch statistics, MishCudaFunction.backward,
 * mmdet/ops/mish_cuda/mish.py:27-36) for the blocks of mmdet/models/backbones/darknetcsp.py.
 *
 * yv4_conv_wgrad: dW[co][(kh,kw,ci)] += sum_m dY[m][co] * x[n, ho*s-p+kh, wo*s-p+kw, ci];
 *   `d` is the FORWARD descriptor; the y_* fields describe the dY view; dW has the packed
 *   forward layout (Cout, KH*KW*Cin) and must be zero on entry (partial sums are added with
 *   float atomics).  Cin, Cout and all view strides/offsets must be multiples of 4.
 * The data gradient is yv4_conv_bn_act_fwd on dY with the weights transposed+flipped (pad
 *   K-1-p); for stride 2 dY is zero-dilated first:
 * yv4_dilate2_fwd: dst (N,2H,2W,C dense) [n,2y,2x,c] = src[n,y,x,c], zeros elsewhere.
 * yv4_bn_train_stats: per-channel batch mean / 1/sqrt(biased var + eps) of an NHWC view with M
 *   rows; updates running_mean/var (unbiased var, `momentum`) when given. work: 4*C doubles (2*C used unless
 *   deterministic).
 * yv4_bn_act_fwd:  y = act((x-mean)*invstd*gamma+beta) (+ residual).
 * yv4_bn_act_bwd:  from dy (gradient w.r.t. y), the saved conv output x and the batch
 *   statistics: dx, dgamma, dbeta (the activation is recomputed, nothing else is saved).  */
int yv4_conv_wgrad(const yv4_conv_desc* d, const float* x, const float* dy, float* dw, void* stream);
int yv4_dilate2_fwd(const float* src, float* dst, int N, int H, int W, int C, int src_cstride,
                    int src_coff, void* stream);
int yv4_bn_train_stats(const float* x, int64_t M, int C, int x_cstride, int x_coff, float eps,
                       float momentum, double* work, float* mean, float* invstd,
                       float* running_mean, float* running_var, void* stream);
int yv4_bn_act_fwd(const float* x, int x_cstride, int x_coff, const float* mean,
                   const float* invstd, const float* gamma, const float* beta,
                   const float* residual, int r_cstride, int r_coff, float* y, int y_cstride,
                   int y_coff, int64_t M, int C, int act, float slope, void* stream);
int yv4_bn_act_bwd(const float* x, int x_cstride, int x_coff, const float* dy, int dy_cstride,
                   int dy_coff, const float* mean, const float* invstd, const float* gamma,
                   const float* beta, float* dx, int dx_cstride, int dx_coff, float* dgamma,
                   float* dbeta, double* work, int64_t M, int C, int act, float slope, void* stream);

/* ---- 16-bit operand convolution (fp16 / bf16 in, fp32 accumulate) ---------------------
 * The reduced-precision form of yv4_conv_bn_act_fwd (BASELINE configs[2..4]; the reference
 * runs its ConvModules in half under mmcv's wrap_fp16_model / autocast, darknetcsp.py:15-35):
 * x, w, residual and (unless out_dtype == YV4_F32) y are NHWC / packed tensors of `dtype`
 * (YV4_F16 or YV4_BF16); scale/shift stay fp32; the epilogue runs in fp32 and rounds once.
 * Same descriptor as the fp32 entry; channel counts, strides and offsets of the 16-bit views
 * must be multiples of 8 (16-byte pixel chunks), w is (Cout, KH*KW*Cin) with Cin padded to 8.
 * desc->tile: YV4_TILE_AUTO or one of YV4_HTILE_*. */
#define YV4_HTILE_128x128 1
#define YV4_HTILE_128x64 2
#define YV4_HTILE_64x64 3
/* 3x3 / stride 1 / pad 1, Cin % 64 == 0, even Cout >= 64, 16-bit output, even channel strides / offsets
 * (conv3x3_pp_h16.hip): one PERSISTENT 8-wave workgroup per CU in ping-pong, 256 pixels x 128 channels per tile, the
 * three kw taps of a (chunk, kh) share one LDS image of the activations, the LDS-DMA ring runs on across tiles, the
 * epilogue stores channel pairs straight from the accumulators.  Same K order and epilogue expressions as the generic
 * tiles.  (Id 5, the 256 x 64 form of round 2's non-persistent kernel, is gone and refused.) */
#define YV4_HTILE_PP3x3 4
/* 3x3 / stride 1 / pad 1 with Cin % 64 == 0 and Cout % 16 == 0 on WIDE wave tiles (16 PT pixels x 64 channels per wave on
 * v_mfma_f32_16x16x32, PT = 2 / 3 / 4 / 6 / 8; workgroup tiles 256 x 256, 192 x 256, 128 x 256, 384 x 128, 256 x 128, 384 x 64 or 256 x 64 chosen
 * per layer so that whole rounds of CUs are filled; csrc/conv3x3_wide_h16.hip).  Same K order and epilogue expressions
 * as the other tiles; measured bit-identical to them (tests/test_gpu_h16.py::test_wide3x3_matches_generic_bitwise). */
#define YV4_HTILE_W3x3 5
/* ... with the tile shape forced (i = 0..6: 256 x 256, 192 x 256, 128 x 256, 384 x 128, 256 x 128, 384 x 64, 256 x 64): tests, tile sweeps */
#define YV4_HTILE_W3x3_SHAPE(i) (5 + 8 * ((i) + 1))
/* The same wave tiles as a general implicit GEMM (any kernel size / stride / padding with Cin % 64 == 0, Cout % 16 == 0,
 * 16-bit output; a K tile = one (64-channel chunk, tap) gathered with the tap's offset): the stride-2 3x3 layers, the deep
 * / wide 1x1 layers, the scattered classes of stride-2 data gradients (csrc/conv_wide_h16.hip).  Bit-identical to the
 * generic tiles.  YV4_HTILE_WIDE_SHAPE(i), i = 0..4, forces a workgroup tile shape as above. */
#define YV4_HTILE_WIDE 8
#define YV4_HTILE_WIDE_SHAPE(i) (8 + 8 * ((i) + 1))
/* 1x1 / stride 1, Cin <= 256, even Cout >= 16 with a 16-bit output (residual allowed) or any Cout >= 16 with an fp32
 * output (no residual) (conv1x1_ws_h16.hip): one persistent 8-wave workgroup per CU, the weight slab resident in LDS,
 * wave-private rings of 32-pixel strips */
#define YV4_HTILE_WS_1x1 6
/* 3x3 / stride 1 / pad 1 with few channels -- Cin 16, 32 or 64, even Cout in [16, 64], 16-bit output
 * (conv3x3_small_h16.hip): one persistent 8-wave workgroup per CU on 16 x 16 output tiles, the weights resident in LDS,
 * the 18 x 18 input tile double-buffered through LDS-DMA */
#define YV4_HTILE_S3x3 7
int yv4_conv_bn_act_fwd_h16(const yv4_conv_desc* d, int dtype, int out_dtype, const void* x,
                            const void* w, const float* scale1, const float* shift1,
                            const float* scale2, const float* shift2, const void* residual,
                            void* y, void* stream);
int yv4_conv_h16_pick_tile(const yv4_conv_desc* d);
/* Stem of the 16-bit path (3-channel image, 3x3 s1 p1, Cout <= 64): x and w are fp32 exactly as
 * for yv4_conv_bn_act_fwd's stem tile, the arithmetic is fp32, only y is `out_dtype`. */
int yv4_conv_stem_fwd(const yv4_conv_desc* d, const float* x, const float* w,
                      const float* scale1, const float* shift1, void* y, int out_dtype,
                      void* stream);

/* The first two layers of CSPDarknet as one launch for the 16-bit inference plans (stem_down_h16.hip): the stem
 * Conv(3 -> C1, 3x3, s1) + BN + act (darknetcsp.py:357-366) and the next stage's conv_downscale Conv(C1 -> C2, 3x3, s2)
 * + BN + act (darknetcsp.py:290-300), straight from the NCHW fp32 image (N, 3, H, W) to the NHWC 16-bit map
 * (N, ceil(H/2), ceil(W/2), C2) at channel offset y_coff of pixels of y_cstride channels.  w1: the fp32 stem weights
 * as yv4_conv_stem_fwd takes them, (C1, 9 taps x 4 channels); the stem is computed on 16-bit MFMAs with a three-term
 * hi / lo split of the fp32 products (result = the fp32 stem's to fp32 accumulation noise) and rounded to `dtype`
 * like its stored output; w2: (C2, 9 * C1) in `dtype`, as yv4_conv_bn_act_fwd_h16 takes it.  C1 in {16, 32},
 * C2 in {32, 64}. */
int yv4_stem_down_fwd_h16(int dtype, const float* x_nchw, int N, int H, int W, const float* w1,
                          const float* scale1, const float* shift1, int C1, int act1, float slope1,
                          const void* w2, const float* scale2, const float* shift2, int C2, int act2,
                          float slope2, void* y, int y_cstride, int y_coff, void* stream);

/* 16-bit forms of the layout adaptors and the SPP pools (same argument meaning as the fp32
 * entries; the boundary tensors stay fp32 NCHW like the reference's, the NHWC side is `dtype`).
 * The nearest resample / concat copy has no 16-bit entry: call yv4_resample_nearest_fwd with the
 * channel arguments halved (a 16-bit view with C % 8 == 0 is an fp32 view with C/2 channels). */
int yv4_nchw_to_nhwc_h16(const float* src, void* dst, int N, int C, int H, int W,
                         int dst_cstride, int dst_coff, int zero_pad, int dtype, void* stream);
int yv4_nhwc_to_nchw_h16(const void* src, float* dst, int N, int C, int H, int W,
                         int src_cstride, int src_coff, int dtype, void* stream);
int yv4_spp_pool_fwd_h16(void* buf, int N, int H, int W, int C, int cstride, int coff,
                         int dtype, void* stream);

/* 16-bit forms of the training kernels (BASELINE configs[2], [4]: bf16 training): activations and
 * their gradients are `dtype` (YV4_F16 / YV4_BF16) NHWC views, statistics / gamma / beta / dW stay
 * fp32, reductions run in double.  Same argument meaning as the fp32 entries above.  The data
 * gradient is yv4_conv_bn_act_fwd_h16 on dY; zero-dilation of a 16-bit map is yv4_dilate2_fwd
 * with the channel arguments halved. */
int yv4_conv_wgrad_h16(const yv4_conv_desc* d, int dtype, const void* x, const void* dy,
                       float* dw, void* stream);
/* Deterministic weight gradient (round 2): same contract as yv4_conv_wgrad / yv4_conv_wgrad_h16 (dw += the weight
 * gradient, dtype YV4_F32 / YV4_F16 / YV4_BF16 operands), but the chunks of the N*Ho*Wo reduction store their partial
 * sums to slabs of `workspace` (yv4_conv_wgrad_workspace bytes; 0 = a single chunk, no workspace needed) and a small
 * kernel adds the slabs to dw in chunk order: run-to-run bit-identical, no float atomics. */
size_t yv4_conv_wgrad_workspace(const yv4_conv_desc* d, int dtype);
int yv4_conv_wgrad_det(const yv4_conv_desc* d, int dtype, const void* x, const void* dy, float* dw,
                       float* workspace, size_t workspace_bytes, void* stream);
int yv4_bn_train_stats_h16(const void* x, int dtype, int64_t M, int C, int x_cstride, int x_coff,
                           float eps, float momentum, double* work, float* mean, float* invstd,
                           float* running_mean, float* running_var, void* stream);
int yv4_bn_act_fwd_h16(const void* x, int dtype, int x_cstride, int x_coff, const float* mean,
                       const float* invstd, const float* gamma, const float* beta,
                       const void* residual, int r_cstride, int r_coff, void* y, int y_cstride,
                       int y_coff, int64_t M, int C, int act, float slope, void* stream);
int yv4_bn_act_bwd_h16(const void* x, int dtype, int x_cstride, int x_coff, const void* dy,
                       int dy_cstride, int dy_coff, const float* mean, const float* invstd,
                       const float* gamma, const float* beta, void* dx, int dx_cstride,
                       int dx_coff, float* dgamma, float* dbeta, double* work, int64_t M, int C,
                       int act, float slope, void* stream);

/* yv4_bn_act_bwd_h16 (eval_mode = 0) / yv4_bn_eval_act_bwd (eval_mode = 1) with dgamma / dbeta ADDED to the given
 * arrays -- the parameters' own gradients -- instead of overwriting them (flags bit 0 = eval_mode). */
int yv4_bn_act_bwd_accum(const void* x, int dtype, int x_cstride, int x_coff, const void* dy,
                         int dy_cstride, int dy_coff, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, void* dx, int dx_cstride,
                         int dx_coff, float* dgamma, float* dbeta, double* work, int64_t M, int C,
                         int act, float slope, int flags /* 1: eval-mode BN, 2: work already zero */,
                         void* stream);

/* SyncBN (the configs under configs/yolov5_ddp: norm_cfg type 'SyncBN' = torch.nn.SyncBatchNorm): the train-mode BN
 * kernels above with the cross-rank exchange between their two halves.  Forward: yv4_bn_partial_sums
 * leaves [sum x (C) | sum x^2 (C)] of the local rows in `work` (double); the caller all-reduces `work`
 * (SUM) and the row count, then yv4_bn_finalize (replicas = 1; > 1: `work` is that many consecutive
 * [2*C] blocks to be added up first, see yv4_conv_fwd_stats) turns the totals into mean / invstd and updates the
 * running statistics (unbiased variance over M_total).  Backward: yv4_bn_act_bwd_sums leaves
 * [sum dz (C) | sum dz*xhat (C)] in `work` and writes the LOCAL dgamma / dbeta (what
 * torch.nn.SyncBatchNorm returns: DDP averages them afterwards); the caller all-reduces `work`;
 * yv4_bn_act_bwd_apply computes dx with the totals over M_total rows.  rows_dev (optional): the total
 * row count as one double in device memory -- it is all-reduced together with the sums, so no host
 * round trip is needed to learn it; when given it overrides M_total. */
int yv4_bn_partial_sums(const void* x, int dtype, int64_t M, int C, int x_cstride, int x_coff,
                        double* work, void* stream);
/* The statistics pass fused into the producing convolution: y = conv(x, w) with the identity epilogue
 * (`ones` / `zeros`: Cout unit scales / zero shifts; y may be a channel slice of a wider buffer) and
 * stats = YV4_STATS_REPLICAS x [sum y (Cout) | sum y^2 (Cout)] doubles whose column sums are the
 * per-channel totals (the kernel spreads its atomics over the replicas; the buffer is cleared here).
 * dtype YV4_F32 / YV4_F16 / YV4_BF16 = type of x, w and y.  Feed it to yv4_bn_finalize with
 * replicas = YV4_STATS_REPLICAS. */
#define YV4_STATS_REPLICAS 64
int yv4_conv_fwd_stats(const yv4_conv_desc* d, int dtype, const void* x, const void* w,
                       const float* ones, const float* zeros, void* y, double* stats,
                       int stats_is_zero /* the caller keeps the buffer clean (yv4_bn_finalize clear_work) */,
                       void* stream);
int yv4_bn_finalize(double* work, int replicas, int64_t M_total, const double* rows_dev, int C,
                    float eps, float momentum, float* mean, float* invstd, float* running_mean,
                    float* running_var, int clear_work /* zero `work` once read */,
                    double* zero_after /* NULL, or 4*C doubles to clear for the layer's backward reduction */,
                    void* stream);
/* The totals of a yv4_conv_fwd_stats buffer as 2*C doubles [sum | sum of squares] (what SyncBN all-reduces before
 * yv4_bn_finalize(replicas = 1)), in either mode; clear_stats: the replicas are zeroed as they are read. */
int yv4_conv_stats_fold(double* stats, int C, int clear_stats, double* out, void* stream);
int yv4_bn_act_bwd_sums(const void* x, int dtype, int x_cstride, int x_coff, const void* dy,
                        int dy_cstride, int dy_coff, const float* mean, const float* invstd,
                        const float* gamma, const float* beta, float* dgamma, float* dbeta,
                        double* work, int64_t M, int C, int act, float slope, void* stream);
int yv4_bn_act_bwd_apply(const void* x, int dtype, int x_cstride, int x_coff, const void* dy,
                         int dy_cstride, int dy_coff, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, void* dx, int dx_cstride,
                         int dx_coff, const double* work, int64_t M, int64_t M_total,
                         const double* rows_dev, int C, int act, float slope, void* stream);

/* A conv weight (Cout, Cin, KH, KW), addressed through its element strides (contiguous or channels_last), to
 * the packed operand of the conv kernels in one pass: rows x (KHo*KWo*ICp), K ordered (kh, kw, channel), the
 * channel count zero-padded to a multiple of pad_to, cast to `dtype`.  Output tap (kh, kw) reads source tap
 * (kh0 + kh*kh_step, kw0 + kw*kw_step).  transpose = 0: rows = Cout, channels = Cin (the forward operand:
 * KHo = KH, kh0 = 0, step 1); 1: rows = Cin, channels = Cout -- mirrored taps (kh0 = KH-1, step -1) give the
 * operand of the data gradient, a tap subset the operand of one parity class of a stride-2 data gradient. */
int yv4_pack_weight(const float* w, int64_t s_co, int64_t s_ci, int64_t s_kh, int64_t s_kw, int Cout,
                    int Cin, int KH, int KW, int KHo, int KWo, int kh0, int kh_step, int kw0,
                    int kw_step, int transpose, int pad_to, void* dst, int dtype, void* stream);

/* yv4_pack_weight over a table of weights in ONE launch (a training step packs every conv weight twice -- forward
 * operand and data-gradient operand -- 750 launches of a few microseconds each on YOLOv4-L; the table is built once and
 * replayed after every optimizer step).  The table lives in DEVICE memory; descriptor i serves workgroups
 * [first_block, first_block + nblocks), rows_per_block output rows each (the caller lays the ranges out back to
 * back, total_blocks = their sum); the other fields are yv4_pack_weight's arguments.  Any rows_per_block >= 1 is
 * correct; the pass is fastest when a workgroup owns whole rows r with all their KHo*KWo taps (forward operand) or groups
 * of max(8, 64 / taps) such rows (transpose = 1: the source is contiguous ACROSS r), since a workgroup stages the source
 * box of its rows in LDS in source order. */
typedef struct yv4_pack_desc {
  const float* w;
  int64_t s_co, s_ci, s_kh, s_kw;
  void* dst;
  int32_t Cout, Cin, KHo, KWo, kh0, kh_step, kw0, kw_step, transpose, pad_to, dtype;
  int32_t first_block, nblocks, rows_per_block;
} yv4_pack_desc;
int yv4_pack_weights_multi(const yv4_pack_desc* table_dev, int n, int total_blocks, void* stream);

/* Backward of yv4_resample_nearest_fwd for INTEGER scale factors (the neck's 2x upsample into its concat buffer,
 * necks/yolo_neck_csp.py:213-219 / torch's upsample_nearest2d_backward): dx (N, Hs, Ws, C) dense = the sum of the
 * (Hd / Hs) x (Wd / Ws) pixels of dy that read it; dy is a channel slice (dy_cstride, dy_coff) of an (N, Hd, Wd, .) tensor
 * of `dtype`; fp32 sum, one rounding. */
int yv4_resample_nearest_bwd(const void* dy, void* dx, int N, int Hs, int Ws, int Hd, int Wd, int C,
                             int dy_cstride, int dy_coff, int dtype, void* stream);

/* One parity class of the data gradient of a stride-2 convolution: a stride-1 convolution of dY whose
 * output pixel (n, ho, wo) is stored at y[n, ho*sh + oh, wo*sw + ow, y_coff + c] of an
 * (N, Hy, Wy, y_cstride) tensor.  d->Ho / d->Wo are taken as given (rows past the input's edge read
 * zeros), d->stride must be 1, the epilogue is scale1/shift1 only.  For the 3x3 / stride 2 / pad 1 convs
 * of darknetcsp.py:288-335 and yolo_neck_csp.py the four classes (1, 2, 2 and 4 taps) cost exactly the
 * forward FLOPs; the zero-dilated form (yv4_dilate2_fwd + yv4_conv_bn_act_fwd) costs 4x. */
int yv4_conv_scatter_fwd(const yv4_conv_desc* d, const float* x, const float* w, const float* scale1,
                         const float* shift1, float* y, int Hy, int Wy, int sh, int sw, int oh,
                         int ow, void* stream);
int yv4_conv_scatter_fwd_h16(const yv4_conv_desc* d, int dtype, const void* x, const void* w,
                             const float* scale1, const float* shift1, void* y, int Hy, int Wy,
                             int sh, int sw, int oh, int ow, void* stream);

/* Backward of an EVAL-mode BatchNorm (+ activation) inside a training graph (frozen stages /
 * norm_eval, darknetcsp.py:466-480): mean / invstd are the running statistics (constants), so
 * dx = gamma * invstd * dy * act'(z); dgamma / dbeta as in yv4_bn_act_bwd.  The forward is
 * yv4_bn_act_fwd(_h16) called with the running statistics. */
int yv4_bn_eval_act_bwd(const void* x, int dtype, int x_cstride, int x_coff, const void* dy,
                        int dy_cstride, int dy_coff, const float* mean, const float* invstd,
                        const float* gamma, const float* beta, void* dx, int dx_cstride,
                        int dx_coff, float* dgamma, float* dbeta, double* work, int64_t M, int C,
                        int act, float slope, void* stream);

/* SPP backward: xcat is the forward's concat buffer (its first C channels are the pooled input),
 * dcat the gradient w.r.t. the 4C-channel concat; dx (N, H, W, C) fp32, dense, ZERO on entry, receives
 * the identity branch plus the three max-pool scatters (first maximum in row-major window order,
 * like ATen).  dtype is the element type of xcat / dcat (YV4_F32 / F16 / BF16). */
int yv4_spp_pool_bwd(const void* xcat, int x_cstride, int x_coff, const void* dcat, int d_cstride,
                     int d_coff, float* dx, int N, int H, int W, int C, int dtype, void* stream);

/* ---- optimizer side of the training step (flat fp32 arenas) -------------------------
 * The reference steps torch.optim.SGD(nesterov) with one param group per parameter
 * (core/custom_hooks/warmup_hooks.py:24-32 requires that), un-scales and clips gradients in
 * Fp16GradAccumulateOptimizerHook.after_train_iter (core/custom_hooks/accum_optim_hooks.py:37-60)
 * and averages all 658 state entries in a Python loop (StateEMAHook.after_train_iter,
 * core/custom_hooks/ema_hooks.py:80-98).  Here parameters, gradients, momentum buffers and EMA
 * copies are slices of four flat arenas and each of those steps is one streaming kernel; every
 * decision (clip coefficient, skip on overflow, loss-scale growth) stays on the device.
 *
 * yv4_grad_prepare: GradScaler.unscale_ + clip_grad_norm_(max_norm, 2) folded into one
 *   multiplier.  scale_state = {scale, growth_tracker} on the device or NULL (scale 1);
 *   max_norm <= 0 disables clipping; work: 2 + YV4_GRAD_PREPARE_MAX_WG doubles (2 used unless deterministic: then
 *   one partial per workgroup, added in index order); ctrl (4 floats, device) receives
 *   {grad multiplier = clip_coef/scale, total L2 norm of the unscaled gradients,
 *    found_inf (0/1), 1/scale}.
 * yv4_sgd_step: p -= lr*(nesterov ? g + m*b : b), b = m*b + g, g = grad*ctrl[0] + wd*p, with
 *   (lr, momentum, weight_decay, nesterov) per segment: seg_off has nseg+1 entries (float
 *   offsets, multiples of 4, seg_off[0] = 0, seg_off[nseg] = n), seg_hyper nseg*4 floats.
 *   No-op when ctrl[2] != 0 (GradScaler.step semantics).  ctrl may be NULL (multiplier 1).
 * yv4_loss_scale_update: GradScaler.update for the dynamic loss scale.
 * yv4_ema_update: ema = momentum*ema + (1-momentum)*online over n floats.                */
#define YV4_GRAD_PREPARE_MAX_WG 2048
int yv4_grad_prepare(const float* grad, int64_t n, const float* scale_state, float max_norm,
                     double* work, float* ctrl, void* stream);
int yv4_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n,
                 const int64_t* seg_off, const float* seg_hyper, int nseg, const float* ctrl,
                 void* stream);
int yv4_loss_scale_update(float* scale_state, const float* ctrl, float growth_factor,
                          float backoff_factor, int growth_interval, void* stream);
int yv4_ema_update(float* ema, const float* online, int64_t n, float momentum, void* stream);

/* ---- YOLOCSPHead training loss, forward and backward ------------------------------------------
 * Replaces responsible_indices (core/anchor/yolov4_anchor_generator.py:12-134, neighbor = 2),
 * get_targets_no_assigner + loss_single_no_assigner (models/dense_heads/yolocsp_head.py:437-575), the
 * decode (yolov4_bbox_coder.py:39-67), aligned GIoU and the three sigmoid-BCE / GIoU reductions for ALL
 * levels: forward = assign + positives + dense objectness pass, backward = positives + one pass that
 * writes the whole gradient of each level's head-conv output (its dtype, its NHWC layout) and the bias
 * gradient.  Nothing returns to the host in between.
 *   raw:   (N, H, W, Cp) head conv output WITHOUT bias, channel c = a*(5+C) + j, Cp >= A*(5+C)
 *   gt (G,4) fp32 x1 y1 x2 y2, gt_label (G) int64, gt_img (G) int64: the batch's ground truths concatenated
 *   candidate slot s = (k*A + a)*G + g  (k: 0 own cell, 1 left, 2 up, 3 right, 4 down) -- the position the
 *   positive has in the reference's index lists; duplicates of one anchor box resolve to the largest s
 *   (what the reference's index_put gives when run in order), deterministically.
 * work buffers (device): slot_anchor, conf_t: L*5*A*G; winner: N*anchors-per-image int32; npos: L int32;
 *   sums: L*3 double = [sum of class BCE | sum of objectness BCE | sum of (1 - GIoU)] per level, in a buffer of
 *   2*L*3 doubles; gpos (backward): L*5*A*G*(5+C) float in a buffer of 4x that many floats; dbias: A*(5+C)
 *   doubles per level in a buffer of twice as many (the second halves are the deterministic mode's lo words;
 *   results are always the doubles / floats at the front).
 * Losses: cls = w_cls*sum/(npos*C), conf = w_conf*sum/(N*H*W*A), bbox = w_bbox*sum/npos (0 where a level has no positive) --
 *   written to `losses` (L*3 float, [cls | conf | bbox] per level) by the forward call itself when the pointer is given
 *   (ABI 6: what yolocsp_head.py:553-575 computes with a dozen tensor ops per level), else left to the host.
 * yv4_yolo_loss_bwd: grad_out (L,3) float (device) = upstream gradients of [cls, conf, bbox] per level. */
#define YV4_LOSS_MAX_LEVELS 5
typedef struct yv4_loss_level {
  const void* raw;
  void* draw;          /* backward: gradient of raw, same shape / dtype */
  const float* bias;   /* A*(5+C) */
  double* dbias;       /* backward: A*(5+C) results, 2*A*(5+C) doubles of room */
  int32_t H, W, Cp, stride;
  float base_anchors[8][4];
} yv4_loss_level;
typedef struct yv4_loss_desc {
  yv4_loss_level levels[YV4_LOSS_MAX_LEVELS];
  int32_t num_levels, N, A, num_classes /* 0: class-agnostic */, G, dtype;
  const float* gt;
  const int64_t* gt_label;
  const int64_t* gt_img;
  float shape_thr, smooth /* one_hot_smoother */, ratio /* conf_iou_loss_ratio */, eps /* GIoU */;
  float w_cls, w_conf, w_bbox;
  int32_t* slot_anchor;
  int32_t* winner;
  int32_t* npos;
  float* conf_t;
  float* gpos;
  double* sums;
  float* losses;       /* optional (ABI 6): L*3 float results of the forward, see above */
} yv4_loss_desc;
int yv4_yolo_loss_fwd(const yv4_loss_desc* d, void* stream);
int yv4_yolo_loss_bwd(const yv4_loss_desc* d, const float* grad_out, void* stream);

/* ---- test-time input pipeline (Resize keep_ratio -> Pad -> Normalize -> ImageToTensor of
 * configs/yolov4/yolov4l_coco_mosaic.py:70-84) for one 8-bit HWC image: bilinear resize with OpenCV's 8-bit
 * fixed-point INTER_LINEAR arithmetic to (new_h, new_w), padding to (Hp, Wp) with pad_val, (v - mean) * (1/std)
 * in fp32 with the channel order swapped if to_rgb (mean / std in OUTPUT channel order, host pointers), written as
 * 3 fp32 planes of Hp*Wp (plane_stride elements apart) -- the image's slot of an NCHW batch.
 * pad_before_normalize: the Pad transform precedes Normalize (the pad value is normalised as well).
 * PARITY UNPINNED: the arithmetic being mirrored is mmcv's / OpenCV's (absent from the build image). */
int yv4_letterbox_u8(const uint8_t* src, int src_h, int src_w, int src_pitch, float* dst, int Hp, int Wp,
                     int64_t plane_stride, int new_h, int new_w, const float* mean3, const float* std3,
                     int to_rgb, int pad_val, int pad_before_normalize, void* stream);

/* ---- evaluation: the reference's two Cython ops, batched over (image, class) problems ----------
 * mmdet/ops/eval_utils/iou/iou_coco.pyx:8-56 and match/match_coco.pyx:8-57, called per image and class
 * from core/evaluation/mean_ap_flexible.py:19-37.  Problem p owns detections [det_off[p], det_off[p+1])
 * (rows of `det`, x1 y1 x2 y2), ground truths [gt_off[p], gt_off[p+1]) and the row-major
 * (num_det x num_gt) IoU block at iou_off[p]; the offset tables have P+1 int64 entries (device).
 * yv4_iou_coco_batched: IoU with the crowd convention (union = detection area for crowd gts), 0 for
 *   non-overlapping pairs, union <= 0 -> 1e-7; bit-exact fp32.
 * yv4_match_coco_batched: per IoU threshold t, detections in order take the best still-available gt
 *   (crowd gts stay available; a match to a regular gt is not given up for an ignore gt);
 *   matched[det_off[p]*num_thrs + t*num_det + d] = gt index inside the problem or -1.
 *   work: num_thrs * total_gt bytes. */
int yv4_iou_coco_batched(const float* det, const float* gt, const uint8_t* is_crowd,
                         const int64_t* det_off, const int64_t* gt_off, const int64_t* iou_off,
                         int P, int64_t total_pairs, float* iou, void* stream);
int yv4_match_coco_batched(const float* iou, const int64_t* det_off, const int64_t* gt_off,
                           const int64_t* iou_off, const float* iou_thrs, int num_thrs,
                           const uint8_t* is_ignore, const uint8_t* is_crowd, int P, uint8_t* work,
                           int32_t* matched, void* stream);

/* ---- split-K form of yv4_conv_bn_act_fwd for single-image (latency) plans ---------------------------------------
 * The reference's only published protocol is batch 1 (tools/analysis_tools/benchmark.py:83-109).  There the deep layers
 * have a handful of output tiles and hundreds of K slices each; this entry splits K over several workgroups per tile
 * (partials in per-split slabs of `workspace`, added in slab order by a finishing kernel that applies the epilogue:
 * deterministic, no atomics).  The summation order differs from yv4_conv_bn_act_fwd's, so the two agree to fp32
 * rounding, not bit for bit.  yv4_conv_splitk_workspace returns the bytes `workspace` must hold (0 and *ksplit = 1
 * when the layer is not split: the call then forwards to yv4_conv_bn_act_fwd). */
size_t yv4_conv_splitk_workspace(const yv4_conv_desc* d, int* ksplit);
int yv4_conv_bn_act_fwd_splitk(const yv4_conv_desc* d, const float* x, const float* w, const float* scale1,
                               const float* shift1, const float* scale2, const float* shift2,
                               const float* residual, float* y, float* workspace, size_t workspace_bytes,
                               void* stream);

/* The same for 16-bit operands (yv4_conv_bn_act_fwd_h16): 64 x 64 tiles, K slices of 64, fp32 partial slabs, a finishing
 * kernel with the 16-bit tiles' epilogue expressions (16-bit or fp32 output).  Layers outside the uniform-K tiles
 * (Cin % 64 != 0) are not split: the call forwards to yv4_conv_bn_act_fwd_h16. */
size_t yv4_conv_h16_splitk_workspace(const yv4_conv_desc* d, int* ksplit);
int yv4_conv_bn_act_fwd_h16_splitk(const yv4_conv_desc* d, int dtype, int out_dtype, const void* x, const void* w,
                                   const float* scale1, const float* shift1, const float* scale2,
                                   const float* shift2, const void* residual, void* y, float* workspace,
                                   size_t workspace_bytes, void* stream);

/* ---- train-side input pipeline (configs/yolov4/yolov4l_coco_mosaic.py:22-69) ----------------------------------
 * Replaces, per batch: Resize(keep_ratio) of 4 source images + MosaicPipeline (mmdet/datasets/pipelines/
 * transforms.py:1906-1983) + the Albu block [PadIfNeeded, RandomCrop, RandomScale, CenterCrop, HorizontalFlip]
 * (albumentations, third party) + HueSaturationValueJitter (transforms.py:1986-2021) + GtBBoxesFilter
 * (transforms.py:2024-2052) + Normalize + ImageToTensor/collate, which the reference runs in CPU dataloader workers.
 * One yv4_aug_image per OUTPUT image describes its four sources and the random draws (made by the host):
 *   src[i], sh/sw/pitch[i]  decoded 8-bit BGR source i (HWC, 3 channels, device memory), i = mosaic tile
 *                           (0 top-left, 1 top-right, 2 bottom-left, 3 bottom-right, transforms.py:1943-1951)
 *   rh/rw[i]                its size after Resize (mmcv rescale_size)
 *   cxy                     mosaic centre = max(rh[0], rh[1], rw[0], rw[2]) (transforms.py:1934); canvas is 2cxy x 2cxy
 *   left, top               PadIfNeeded offsets; x1, y1 RandomCrop origin in the padded canvas; C crop size (1280)
 *   S                       size of the crop after RandomScale (int(C * scale)); o = (S - out) / 2 CenterCrop origin
 *   flip                    HorizontalFlip applied; hsv_on + lut[3][256]: the hue / saturation / value LUTs of
 *                           transforms.py:2004-2008 for this image's random gains. */
typedef struct {
  const void* src[4];
  int32_t sh[4], sw[4], pitch[4];
  int32_t rh[4], rw[4];
  int32_t cxy, left, top, x1, y1, C, S, o, flip, hsv_on;
  uint8_t lut[3][256];
} yv4_aug_image;

/* Pixels of N output images (out_size x out_size).  imgs: N descriptors in DEVICE memory.  out_u8 (optional):
 * (N, out, out, 3) 8-bit BGR image after the geometric chain, before the colour jitter; out_nchw (optional):
 * (N, 3, out, out) fp32 after jitter + (v - mean) * (1 / std) with the BGR -> RGB swap of to_rgb. */
int yv4_mosaic_augment_u8(const yv4_aug_image* imgs, int N, int out_size, uint8_t* out_u8, float* out_nchw,
                          const float* mean3, const float* std3, int to_rgb, int pad_val, void* stream);

/* Boxes of the same N output images.  boxes (total, 4) pascal_voc in SOURCE image coordinates, labels / tile (total,)
 * (tile = which of the 4 sources the box belongs to), seg (N + 1) ranges.  Per box: Resize scale + clip (float32),
 * mosaic shift, the Albu chain in float64 with its BboxParams filter (clipped area > min_area, clipped / unclipped
 * area > min_visibility), GtBBoxesFilter (w, h > min_size, aspect ratio < max_aspect_ratio).  Survivors are written in
 * input order to out_boxes (N, cap, 4) / out_labels (N, cap), their number (<= cap) to out_count (N). */
int yv4_augment_boxes(const yv4_aug_image* imgs, int N, int out_size, const float* boxes, const int32_t* labels,
                      const int32_t* tile, const int64_t* seg, int cap, double min_area, double min_visibility,
                      float min_size, float max_aspect_ratio, float* out_boxes, int32_t* out_labels,
                      int32_t* out_count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YV4_H_ */
