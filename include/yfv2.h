/* yfv2.h - C ABI of libyfv2.so: the MI355X-native (gfx950) Yolo-FastestV2 forward
 * detection path.  This header is the drop-in boundary: plain C, raw device
 * pointers and sizes, no torch / C++ types.  The reference (dog-qiuqiu/
 * Yolo-FastestV2) is pure Python and has no FFI of its own; each entry point
 * below replaces the named reference Python symbol (file:line under the
 * reference tree), and yolo_fastestv2_amd/ binds them with ctypes
 * (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - every function returns 0 on success or a negative yfv2_status; nothing
 *     throws across the ABI; yfv2_last_error() gives the message for a handle
 *     (or for the calling thread when the handle is NULL / creation failed).
 *   - the caller owns every buffer it passes; the library owns only its weight
 *     blob and workspace (allocated in yfv2_create for `max_batch` images).
 *   - all work is enqueued on the caller's HIP stream (`stream` is a
 *     hipStream_t passed as void*; NULL = the default stream).  No hidden
 *     device synchronisation, except in yfv2_profile_forward and
 *     yfv2_debug_activation, which are measurement/debug helpers and say so.
 *   - all pointers named x / out6 / boxes / dets / idx / count are DEVICE
 *     pointers; weights passed to yfv2_load_weights are HOST pointers.
 *     (yfv2_plan.lanes = N at yfv2_create_ex: a forward / detect of a large batch is cut into N slices that run on
 *     N streams owned by the handle - forked from and joined back into the caller's stream with events inside the call, same
 *     ordering contract, bit-identical results; DESIGN.md section 5.  Memory: the parent handle keeps its full workspace
 *     for max_batch images - it serves batches below the slicing threshold, yfv2_profile_forward and the training entry
 *     points - and every lane adds a workspace for max_batch / N images plus its own copy of the 1 MB weight blob:
 *     about twice the activation memory of a handle without lanes, ~3 GB at max_batch 256.)
 *   - one handle per device, not thread-safe, and ONE STREAM AT A TIME: the handle's workspace (activations, logits
 *     and candidate rows of yfv2_detect, the class-filter scratch of yfv2_nms) is shared by all calls on it, so calls
 *     issued on different streams must be ordered by the caller (events); use one handle per concurrent stream.
 *   - there is no CPU fallback: if no gfx950 device is usable, yfv2_create
 *     fails with YFV2_ERR_DEVICE.
 */
#ifndef YFV2_H
#define YFV2_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YFV2_ABI_VERSION 6 /* 2: yfv2_stage_info reports external bytes as well; 3: yfv2_train_*, yfv2_sgd_step; 4: yfv2_nonfinite, lanes; 5: yfv2_nonfinite_peek, yfv2_clock_probe_*; 6: yfv2_plan / yfv2_create_ex (the library reads no environment variable) */
#define YFV2_API __attribute__((visibility("default")))
#define YFV2_MAX_DET 300 /* utils/utils.py:243 max_det */

typedef enum yfv2_status {
  YFV2_OK = 0,
  YFV2_ERR_ARG = -1,     /* NULL / out-of-range argument */
  YFV2_ERR_CONFIG = -2,  /* unsupported model configuration */
  YFV2_ERR_DEVICE = -3,  /* no usable gfx950 device / HIP runtime error */
  YFV2_ERR_WEIGHTS = -4, /* missing / mis-shaped tensor in yfv2_load_weights */
  YFV2_ERR_STATE = -5,   /* call order (e.g. forward before load_weights) */
  YFV2_ERR_BATCH = -6,   /* B < 1 or B > max_batch */
  YFV2_ERR_RANGE = -7    /* reported by the HOST classes (include/yfv2.hpp detection(), the Python Engine): the range guard of the
                            default plan tripped (yfv2_nonfinite) - the results of that call are invalid */
} yfv2_status;

typedef struct yfv2_ctx* yfv2_handle;

/* Mirrors the cfg dict the reference reads from its ".data" file
 * (utils/utils.py:13-65: classes, anchor_num, width, height, anchors) plus
 * the workspace size. */
typedef struct yfv2_config {
  int32_t classes;     /* 80; 1..255 */
  int32_t anchor_num;  /* 3 (the reference hard-codes 3: utils/utils.py:300,326) */
  int32_t height;      /* 352; multiple of 32 */
  int32_t width;       /* 352; multiple of 32; at most 4096 decode rows = 3 (H/16 W/16 + H/32 W/32), i.e. up to 512x512 */
  double anchors[12];  /* data/coco.data:17, float64 like utils/utils.py:305 */
  int32_t max_batch;   /* workspace is sized for this many images */
  int32_t device;      /* HIP device ordinal */
} yfv2_config;

/* One named fp32 tensor of a reference state_dict (host memory). */
typedef struct yfv2_tensor_desc {
  const char* name;  /* reference key, e.g. "backbone.stage2.0.branch_main.0.weight" */
  const float* data; /* host pointer, contiguous fp32 */
  int64_t numel;
} yfv2_tensor_desc;

/* Plan switches of a handle: WHICH kernels compute the path, never WHAT it computes (every plan is held to the same parity
 * tests, tests/test_gpu_parity.py::test_fallback_plans_match_oracle).  All zero = the default plan.  The library reads no
 * environment variable: this struct, passed to yfv2_create_ex, is the only way a plan is chosen (the Python layer and the
 * tools map YFV2_BF6=0 / YFV2_FUSED=0 / YFV2_POSTFUSE=0 / YFV2_FRONT=0 / YFV2_TPAIR=0 / YFV2_LANES=N / YFV2_TRACE=1 of THEIR
 * environment onto it, yolo_fastestv2_amd/_lib.py plan_from_env). */
typedef struct yfv2_plan {
  int32_t struct_size;        /* sizeof(yfv2_plan) of the caller's header (the struct may grow at its end) */
  int32_t fp32_matrix;        /* 1: every channel contraction on the fp32 matrix instructions (the reference's own arithmetic: no
                                 range limit, about half the speed) instead of fp16x3 on the f16 matrix cores */
  int32_t layer_by_layer;     /* 1: one launch per reference layer (77 launches), NHWC throughout: the general plan every fused
                                 kernel was first validated against */
  int32_t post_two_launches;  /* 1: yfv2_detect's decode and NMS as two launches (the only form beyond 96 classes / 2048 rows) */
  int32_t front_two_launches; /* 1: the stem and stage2.0 as two launches */
  int32_t towers_unpaired;    /* 1: the tower halves of the 22x22-class level as four launches instead of two */
  int32_t lanes;              /* N > 1: one call on a large batch is cut into N slices on N streams the handle owns; 0 / 1: off */
  int32_t trace;              /* 1: debug - per-wave cycle stamps of the stamped kernels (yfv2_debug_activation(100)) ... */
  int32_t trace_step;         /* ... of launch `trace_step` of the plan only (-1: of every stamped launch) */
} yfv2_plan;

/* ---- lifetime -------------------------------------------------------------- */

/* replaces: model/detector.py:8-19 Detector.__init__ (+ .to(device)).  yfv2_create(out, cfg) = yfv2_create_ex(out, cfg, NULL):
 * the default plan. */
YFV2_API int yfv2_create(yfv2_handle* out, const yfv2_config* cfg);
YFV2_API int yfv2_create_ex(yfv2_handle* out, const yfv2_config* cfg, const yfv2_plan* plan);
YFV2_API void yfv2_destroy(yfv2_handle h);
YFV2_API const char* yfv2_last_error(yfv2_handle h);
YFV2_API int yfv2_abi_version(void);

/* replaces: nn.Module.load_state_dict (test.py:28, evaluation.py:53).  Takes the
 * reference key set (BN running stats included, num_batches_tracked ignored),
 * folds BN to per-channel scale/shift (eval mode, eps 1e-5), re-lays-out the
 * filters for the kernels and uploads them.  Synchronous. */
YFV2_API int yfv2_load_weights(yfv2_handle h, const yfv2_tensor_desc* tensors, int32_t n);

/* Replace the 6 anchor pairs used by yfv2_decode / yfv2_detect (the reference passes
 * cfg["anchors"] to handel_preds on every call, utils/utils.py:305-306). Host-side only. */
YFV2_API int yfv2_set_anchors(yfv2_handle h, const double anchors[12]);

/* ---- the hot path ---------------------------------------------------------- */

/* replaces: model/detector.py:21-47 Detector.forward (export_onnx=False).
 * x: (B,3,H,W) fp32 NCHW in [0,1], 16-byte aligned (YFV2_ERR_ARG otherwise; the uint8 entry points: 4-byte aligned) (test.py:38; any |x| < 255.9 is computed to fp32 accuracy, beyond that the default
 * plan's stem - two-term fp16 operands on the matrix cores, yfv2_stem16.hip - leaves fp16's range: DETECTED, see
 * yfv2_nonfinite below; a handle created with yfv2_plan.fp32_matrix = 1 and the uint8 entry point yfv2_forward_u8 have no such bound).  out6: six NCHW fp32 logit tensors in the
 * reference's return order (reg_2, obj_2, cls_2, reg_3, obj_3, cls_3) with
 * shapes (B,4A,H/16,W/16) (B,A,..) (B,classes,..) and the same at H/32. */
YFV2_API int yfv2_forward(yfv2_handle h, const float* x, int32_t B, float* const out6[6], void* stream);

/* Range guard of the default plan.  Its channel contractions run as "fp16x3" (every fp32 operand split into two fp16 terms
 * after an exact power-of-two scale: fp32-accurate) and are valid while |activation| < 4094 and, for
 * yfv2_forward's fp32 input, |x| < 255.9 - two orders of magnitude above anything the COCO checkpoint or a He-initialised
 * network produces.  Beyond that an operand becomes (+Inf, -Inf), its products NaN, and the ReLU behind the conv would turn
 * the NaN into a silent 0; the reference's fp32 conv has no such cliff.  So every kernel of that plan tests its matrix-core
 * accumulators BEFORE the ReLU and sets a sticky word in the handle when one is not a number (also true for non-finite
 * values in x itself).  yfv2_nonfinite waits for `stream`, writes 1 to *flag if any forward / detect since the last query
 * tripped the guard (0 otherwise) and clears the word.  A handle created with yfv2_plan.fp32_matrix = 1 computes every
 * conv on the fp32 matrix instructions, has no such bound and never sets it.  The Python surface queries the word wherever it
 * synchronises anyway (handel_preds, non_max_suppression's callers, evaluation) and raises. */
YFV2_API int yfv2_nonfinite(yfv2_handle h, int32_t* flag, void* stream);
/* The same word WITHOUT waiting for anything and without clearing it: 1 if a kernel that has already completed tripped the
 * guard.  The word lives in host-mapped memory, so this is a host memory read - free.  For callers that never synchronise
 * with the host between calls (a detect loop that hands device tensors on, include/yfv2.hpp, Engine.detect): they look before
 * every call and learn of a tripped guard one call late instead of never; yfv2_nonfinite (after the results were waited
 * for) stays the exact query and the one that clears. */
YFV2_API int yfv2_nonfinite_peek(yfv2_handle h, int32_t* flag);

/* Same forward from the image layout the reference's callers hold BEFORE their pre-process step
 * (test.py:34-38, utils/datasets.py:106-111): uint8 (B, height, width, 3) - HWC, channel order as decoded
 * (BGR for cv2), values 0..255 - already resized to the configured size.  Replaces
 * `img.reshape(1,H,W,3); torch.from_numpy(img.transpose(0,3,1,2)); img.to(device).float()/255.0` + Detector.forward:
 * the transpose/cast is done by the stem kernel's loads, the 1/255 is folded into its filter (logits agree with
 * the fp32 path to rounding, same 1e-4 bound).  SURVEY.md section 8(f) row 1. */
YFV2_API int yfv2_forward_u8(yfv2_handle h, const uint8_t* x, int32_t B, float* const out6[6], void* stream);

/* replaces: utils/utils.py:303-358 handel_preds (+ make_grid :298-300).
 * boxes: (B, rows, 5+classes) fp32, rows = A*(H/16*W/16 + H/32*W/32) = 1815,
 * row order (y, x, anchor) per scale, scale 0 then 1; columns cx,cy,w,h,obj,cls. */
YFV2_API int yfv2_decode(yfv2_handle h, const float* const out6[6], int32_t B, float* boxes, void* stream);

/* replaces: utils/utils.py:232-296 non_max_suppression incl. xywh2xyxy :67-74
 * and torchvision.ops.nms (called at :286), without the 1 s wall-clock abort.
 * conf_thres is compared in fp32 (as the reference's tensor>python-float does),
 * iou_thres in double (as torchvision's kernel does).  classes may be NULL.
 * dets: (B,300,6) fp32 rows x1,y1,x2,y2,conf,cls, descending conf, first
 * count[b] rows valid; idx: (B,300) int32 = row index of each survivor in the
 * decode order; count: (B) int32. */
YFV2_API int yfv2_nms(yfv2_handle h, const float* boxes, int32_t B, float conf_thres, double iou_thres,
             const int32_t* classes, int32_t n_classes, float* dets, int32_t* idx, int32_t* count,
             void* stream);

/* forward -> decode -> NMS in one call (test.py:42,48,49 / utils/utils.py:379-383).
 * Intermediate logits and compact candidate rows live in the handle's workspace; the
 * (B,rows,5+classes) tensor is not materialised on this path (same arithmetic, same result). */
YFV2_API int yfv2_detect(yfv2_handle h, const float* x, int32_t B, float conf_thres, double iou_thres,
                float* dets, int32_t* idx, int32_t* count, void* stream);

/* yfv2_detect from uint8 (B, height, width, 3) images (see yfv2_forward_u8). */
YFV2_API int yfv2_detect_u8(yfv2_handle h, const uint8_t* x, int32_t B, float conf_thres, double iou_thres,
                            float* dets, int32_t* idx, int32_t* count, void* stream);

/* replaces: cv2.resize(img, (cfg.width, cfg.height), interpolation=cv2.INTER_LINEAR) of test.py:35 and
 * utils/datasets.py:107 for B equally sized uint8 HWC frames: src (B, src_h, src_w, 3) -> dst (B, cfg.height,
 * cfg.width, 3), both on the device, dst 4-byte aligned; dst is what yfv2_forward_u8 / yfv2_detect_u8 take.  The
 * arithmetic is OpenCV's 8-bit fixed-point bilinear path (11-bit coefficients, see yfv2_pre.hip); src size == dst
 * size is an exact copy.  Source rows up to ~27 000 pixels wide.  SURVEY.md section 8(f) row 1. */
YFV2_API int yfv2_resize_u8(yfv2_handle h, const uint8_t* src, int32_t B, int32_t src_h, int32_t src_w, uint8_t* dst, void* stream);

/* replaces: utils/utils.py:194-230 get_batch_statistics (with bbox_iou :76-108), the per-detection loop of
 * evaluation() (:361-395).  dets/count: the padded output of yfv2_nms / yfv2_detect; targets: (T,6) fp32 device rows
 * [image index, label, x1, y1, x2, y2] in pixels, i.e. what evaluation() holds after utils.py:372-376; tp: (B,300)
 * int32, 1 where the reference's true_positives is 1.  Same walk as the reference: detections in their (score-
 * descending) order, best-IoU target over all of the image's targets (first on ties), IoU with the "+1 pixel"
 * convention in fp32, threshold compared in fp32, a target is matched once, stop when all targets are matched.
 * At most 1024 targets per image (YFV2_ERR_ARG otherwise; COCO's maximum is below 100).  B is not bound by max_batch
 * (no workspace is involved).  Unlike the other entry points this one waits for the stream before it returns (it reads
 * the overflow flag back); evaluation loops use the two-call form below instead.  SURVEY.md section 8(f) row 2. */
YFV2_API int yfv2_batch_statistics(yfv2_handle h, const float* dets, const int32_t* count, int32_t B, const float* targets,
                                   int32_t T, float iou_threshold, int32_t* tp, void* stream);
/* The same launch, enqueue only (no host synchronisation): an image with more than 1024 targets sets a sticky flag
 * inside the handle (its own device word, not shared with any other entry point).  yfv2_batch_statistics_overflow
 * waits for `stream`, returns that flag through *overflowed (1: at least one call since the last query produced an
 * invalid tp row) and clears it - call it once after the last batch (utils.evaluation does). */
YFV2_API int yfv2_batch_statistics_async(yfv2_handle h, const float* dets, const int32_t* count, int32_t B, const float* targets,
                                         int32_t T, float iou_threshold, int32_t* tp, void* stream);
YFV2_API int yfv2_batch_statistics_overflow(yfv2_handle h, int32_t* overflowed, void* stream);

/* replaces: utils/loss.py:130-208 compute_loss (with build_target :53-124 and bbox_iou(CIoU) :8-51) and, when grad6 is
 * not NULL, what autograd derives from it: the gradient of the TOTAL loss w.r.t. each of the six logit maps (same
 * shapes as out6; device pointers; overwritten).  SURVEY.md section 8(f) row 3, first slice - the loss end of the
 * training path (train.py:105-108); train-mode forward and the convolutions' backward are not implemented.
 *   out6     the six NCHW logit maps of yfv2_forward (B images)
 *   targets  (T, 6) fp32 device rows [image index in the batch, class, cx, cy, w, h], box normalised to [0, 1]: the tensor
 *            the reference's collate_fn builds (utils/datasets.py:12-22); T may be 0
 *            PRECONDITION: 0 <= image index < B and 0 <= class < classes; a row that violates it is skipped (the
 *            reference's CrossEntropyLoss raises on such a class), it is never used as an index
 *   losses   device float[4]: lbox (x3.2), lobj (x64), lcls (x32) and their sum, the 4-tuple compute_loss returns
 * Uses the handle's anchors (yfv2_set_anchors) as float64 like the reference.  Work is enqueued on `stream`; the
 * handle's loss workspace grows (one device synchronisation) when T exceeds what earlier calls needed. */
YFV2_API int yfv2_loss(yfv2_handle h, const float* const out6[6], int32_t B, const float* targets, int32_t T, float* losses,
                       float* const grad6[6], void* stream);

/* ---- the rest of the training path: one iteration of train.py:96-123 on the device.
 * Parameters, their gradients and the BatchNorm buffers are the CALLER's device tensors in the reference's own layouts and
 * under the reference's state_dict names; nothing is copied.  Correctness-first kernels (plain NCHW fp32, float64
 * reductions), not the throughput path.
 *
 * yfv2_train_bind      tensors: every floating-point entry of the state_dict (weights, biases, running_mean / running_var) as
 *                      DEVICE pointers; grads: one device buffer per trainable parameter (same names).  Pointers must stay
 *                      valid until the next bind.
 * yfv2_train_forward   replaces model/detector.py:21-47 Detector.forward in train() mode (train.py:105): every BatchNorm on
 *                      batch statistics (biased variance), running statistics moved by momentum 0.1 with the unbiased
 *                      variance (num_batches_tracked is the caller's integer); writes the six NCHW logit maps and records
 *                      what the backward needs in a workspace that grows with B.
 * yfv2_train_backward  replaces total_loss.backward() (train.py:110) from the logits down: grad6 = gradient of the loss
 *                      w.r.t. the six logit maps (yfv2_loss); ADDS the gradient of every parameter to its bound buffer (zero
 *                      them first; accumulating over `subdivisions` batches as train.py:122 does is then free).
 * yfv2_sgd_step        replaces optimizer.step() (torch.optim.SGD as train.py:81-85 builds it: momentum, weight_decay,
 *                      dampening 0, no Nesterov) for ONE parameter tensor of n elements: d = g + wd p; buf = first_step ? d :
 *                      momentum buf + d; p -= lr buf.  The warm-up of train.py:113-117 only changes `lr`. */
YFV2_API int yfv2_train_bind(yfv2_handle h, const yfv2_tensor_desc* tensors, int32_t n, const yfv2_tensor_desc* grads, int32_t ng);
YFV2_API int yfv2_train_forward(yfv2_handle h, const float* x, int32_t B, float* const out6[6], void* stream);
YFV2_API int yfv2_train_backward(yfv2_handle h, const float* const grad6[6], void* stream);
YFV2_API int yfv2_sgd_step(yfv2_handle h, float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                           float weight_decay, int32_t first_step, void* stream);
/* The same update for MANY parameter tensors in a few launches (the network has 225: one launch each is 225 kernels of two
 * microseconds and as many gaps).  items: HOST array, one entry per tensor (device pointers inside); the entries travel in the
 * kernel-argument segment, 96 per launch. */
typedef struct yfv2_sgd_item {
  float* param; const float* grad; float* momentum_buf; /* device pointers */
  int64_t n;                                            /* elements */
  int32_t first_step; int32_t reserved;                 /* first_step: momentum_buf is uninitialised (buf = d) */
} yfv2_sgd_item;
YFV2_API int yfv2_sgd_step_multi(yfv2_handle h, const yfv2_sgd_item* items, int32_t n_items, float lr, float momentum, float weight_decay,
                                 void* stream);

/* ---- introspection / measurement (bench.py, tests) ------------------------- */

YFV2_API int32_t yfv2_num_rows(yfv2_handle h);   /* 1815 for 352x352, A=3 */
YFV2_API int32_t yfv2_num_stages(yfv2_handle h); /* launches in one forward */

/* Static description of launch `i` of the forward plan: kernel name, which
 * reference layers it covers, and its ALGORITHMIC work per image: flops =
 * 2*MACs; bytes_per_image = per-LAYER accounting (every reference layer the
 * launch covers reads its input and writes its output once, fp32: BASELINE.md
 * section 4); external_bytes_per_image = SURVEY.md section 8(d)'s figure for a
 * fused launch - only what the launch itself must read from and write to HBM
 * (the chain of seven stride-1 blocks: one read + one write of the activation).
 * bench.py's roofline uses the external figure. */
YFV2_API int yfv2_stage_info(yfv2_handle h, int32_t i, char* name, int32_t name_cap, double* flops_per_image,
                    double* bytes_per_image, double* external_bytes_per_image);
/* The kernel (family) launch `i` runs, as a prefix of the symbol name a rocprofv3 kernel trace shows for it
 * (e.g. "block_s1chain_kernel", "tower2_kernel<6, 512, 4, 4>"): lets bench.py group its per-launch times the way
 * the kernel-stats tables under profiles/ do. */
YFV2_API int yfv2_stage_kernel(yfv2_handle h, int32_t i, char* name, int32_t name_cap);

/* Measurement helper (synchronises once, at the end): one untimed pass, then
 * `iters` passes of the forward queued back to back on `stream`, every launch
 * recording its own begin and end into a hipEvent pair (hipExtLaunchKernel start /
 * stop events: the dispatch's timestamps, as a rocprofv3 trace reports them; a step
 * of several launches: first begin to last end); the mean duration of each step in
 * milliseconds goes to ms[0..num_stages).  Between two passes the decode + NMS launch of
 * yfv2_detect runs untimed (thresholds 0.3 / 0.4, results discarded) where the
 * configuration has the fused form, so that a pass's first launch follows what it
 * follows in a detect loop. */
YFV2_API int yfv2_profile_forward(yfv2_handle h, const float* x, int32_t B, float* const out6[6], int32_t iters,
                         float* ms, void* stream);

/* Measurement helper: the EFFECTIVE shader clock of the device, measured by the shader itself.  `workgroups` one-wave
 * workgroups stay on the device for `milliseconds` and stamp s_memtime (shader cycles) against s_memrealtime (the constant
 * reference clock, hipDeviceAttributeWallClockRate); the quotient is the clock the wave's engine ran at over the interval -
 * DVFS, power cap and performance level included.  busy = 1: dependent FMAs between the stamps (launch >= one workgroup per
 * CU: the clock under an all-CU vector load); busy = 0: the waves sleep between looks at the reference clock and take no issue
 * slots worth naming - launched with a handful of workgroups on a SIDE stream while forwards run on the main stream, they
 * report the clock those forwards' kernels actually ran at.  yfv2_clock_probe_begin only enqueues on `stream`;
 * yfv2_clock_probe_end waits for `stream` and writes out[0..2] = min / mean / max MHz over the workgroups, out[3] = reference
 * clock in MHz, out[4] = mean measured interval in ms, out[5] = number of distinct XCDs the workgroups ran on.  bench.py puts
 * these figures next to every timing it reports (boxes of one pool differ). */
YFV2_API int yfv2_clock_probe_begin(yfv2_handle h, int32_t workgroups, float milliseconds, int32_t busy, void* stream);
YFV2_API int yfv2_clock_probe_end(yfv2_handle h, double out[6], void* stream);

/* Measurement helper: one whole forward, then launch `step` of the plan (0 .. yfv2_num_stages) `iters` times back to back on
 * `stream`; enqueue only.  A launch repeated for a few hundred milliseconds is long enough for the device's power sensor:
 * tools/power_probe.py reads it meanwhile and prices every launch in joules (the pipelined headline is power-limited). */
YFV2_API int yfv2_debug_repeat_step(yfv2_handle h, const float* x, int32_t B, float* const out6[6], int32_t step, int32_t iters, void* stream);

/* Debug/parity helper: copy one internal NHWC activation of the LAST forward
 * to host as (B,H,W,C).  which: 0 stem+pool, 1 stage2, 2 stage3 (C2), 3 stage4
 * (C3), 4 S2 (fpn 22x22), 5 S3 (fpn 11x11).  Returns the element count.
 * which = 0 on the default plan: the stem's output never exists in memory (it is fused into stage2.0's launch), so the hook RE-RUNS
 * the stem's own launch on the input pointer of the last forward / detect - the caller must still hold that buffer, unchanged
 * (YFV2_ERR_STATE if no forward of at least B images has run on the handle since it was created). */
YFV2_API int64_t yfv2_debug_activation(yfv2_handle h, int32_t which, int32_t B, float* host_dst, int64_t cap);

/* Debug/parity helper for the training path: the output of one ReLU'd conv+BatchNorm of the LAST yfv2_train_forward, dense
 * (B,C,H,W), copied to host.  conv_name is the state_dict prefix of its conv ("backbone.stage2.0.branch_main.0",
 * "fpn.cls_head_2.block.5", ...).  Its sign pattern is the set of ReLU decisions that execution took - the gradient parity
 * test replays the float64 oracle on exactly those decisions (tests/test_train_gpu.py).  Returns the element count (with
 * host_dst NULL or capacity too small: the count needed, nothing copied), -1 on error.  Synchronises the device. */
YFV2_API int64_t yfv2_debug_train_relu_output(yfv2_handle h, const char* conv_name, float* host_dst, int64_t capacity);

/* Host-only test hook: validates cfg, builds the launch plan and packs the weights exactly as yfv2_create +
 * yfv2_load_weights do, WITHOUT a device (the workspace gets made-up addresses used only for pointer arithmetic);
 * reports the number of launches and the packed blob size.  Lets the CPU test suite exercise the host logic for every
 * (classes, height, width) the configuration check admits.  Never launches or computes anything. */
YFV2_API int yfv2_debug_plan_dryrun(const yfv2_config* cfg, const yfv2_tensor_desc* tensors, int32_t n, int32_t* n_steps, int64_t* blob_floats);
/* the same for a plan other than the default (NULL = the default plan) */
YFV2_API int yfv2_debug_plan_dryrun_ex(const yfv2_config* cfg, const yfv2_plan* plan, const yfv2_tensor_desc* tensors, int32_t n, int32_t* n_steps,
                                       int64_t* blob_floats);
/* Host-only test hook: the packed LDS image of launch `step` of that plan (up to `cap` floats from the image's start) and
 * the launch's name; returns the number of floats copied or a negative error code.
 * INDEX SPACE: images are packed per reference block, so this hook numbers the steps of the IMAGE VIEW: where the default plan runs
 * the stem and stage2.0 as one launch (front2_kernel) the view still has two steps - 0 = the stem's image, 1 = stage2.0's, launch k of
 * the plan (yfv2_num_stages / yfv2_stage_info / yfv2_profile_forward / yfv2_debug_repeat_step index the LAUNCHES) = view step k + 1.
 * step = -1 returns the number of view steps (dst may be NULL).  step + 1000 (j + 1) = job j of a launch that runs several tower halves. */
YFV2_API int64_t yfv2_debug_plan_image(const yfv2_config* cfg, const yfv2_tensor_desc* tensors, int32_t n, int32_t step, char* name,
                                       int32_t name_cap, float* dst, int64_t cap);
YFV2_API int64_t yfv2_debug_plan_image_ex(const yfv2_config* cfg, const yfv2_plan* plan, const yfv2_tensor_desc* tensors, int32_t n, int32_t step,
                                          char* name, int32_t name_cap, float* dst, int64_t cap);   /* ... of a plan other than the default */
/* Host-only test hook: the channel order in which that plan stores stage 3's output C2 (label[k] = logical channel at
 * NHWC position k, 96 entries); returns 1 if the plan permutes (chain kernel), 0 for plain NHWC, negative on error. */
YFV2_API int yfv2_debug_plan_c2_label(const yfv2_config* cfg, const yfv2_tensor_desc* tensors, int32_t n, int32_t* label);

#ifdef __cplusplus
}
#endif
#endif /* YFV2_H */
