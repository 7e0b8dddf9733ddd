/*
 * hd_ops.h — C ABI of the MI355X (gfx950) detection hot-path library, libhd_ops.so.
 *
 * What this replaces.  Scienthusiasts/HeltonDetection is a pure-PyTorch detector
 * framework (reference README.md:2) whose FasterRCNN (README.md:8) and YOLOv5
 * (README.md:9) heads/assigners/post-processing call PyTorch ops for anchors, IoU,
 * NMS, RoIAlign (README.md:65,73), grid decode, box loss and WBF (README.md:19).
 * The mounted reference snapshot holds NO source (README + two images only,
 * SURVEY.md §0), so no reference signature can be cited beyond those README lines;
 * each entry point below names the README line of the feature it serves and the
 * public algorithm definition it implements.  Parity is therefore pinned to this
 * repo's own CPU restatement under oracle/ ("parity unpinned").
 *
 * Contract:
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every pointer is a DEVICE pointer unless the parameter is documented "host".
 *   - the caller owns all memory; the library never allocates, frees or synchronises.
 *   - every launch is asynchronous on `stream` (a hipStream_t passed as void*).
 *   - scratch memory comes from a caller-provided workspace; each op with a
 *     workspace has a companion *_workspace_bytes() query (pure host arithmetic).
 *   - return value: HD_OK (0) or a negative hd_status; nothing is thrown.
 *   - re-entrant and thread-safe given distinct workspaces/outputs.
 *   - boxes are float32 xyxy in continuous pixel coordinates (no legacy +1).
 *   - all tensors are dense row-major in the shape written next to them unless
 *     explicit element strides are part of the signature.
 *
 * Conventions (IoU formula, tie-breaks, RoIAlign sampling, ...) are frozen in
 * DESIGN.md §2 and restated in oracle/hd_oracle.c, the checker for every function.
 */
#ifndef HD_OPS_H
#define HD_OPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int hd_status;
typedef void* hd_stream; /* hipStream_t */

#define HD_OK 0
#define HD_ERR_NULL (-1)       /* required pointer is NULL            */
#define HD_ERR_SHAPE (-2)      /* a size/shape argument is out of range */
#define HD_ERR_ALIGN (-3)      /* pointer not aligned as documented   */
#define HD_ERR_WORKSPACE (-4)  /* workspace missing or too small      */
#define HD_ERR_LAUNCH (-5)     /* hipLaunch / hipMemsetAsync failed   */
#define HD_ERR_NO_DEVICE (-6)  /* no HIP device visible               */

#define HD_MAX_LEVELS 8   /* pyramid levels one call may carry */
#define HD_MAX_ANCHORS 16 /* anchors per location              */

/* ---- library / device ------------------------------------------------------ */

/* ABI version, bumped on any signature change. */
int hd_abi_version(void);
/* Human-readable status text (static storage). */
const char* hd_status_str(hd_status s);
/* Number of visible HIP devices (0 on a CPU-only host); never throws. */
int hd_device_count(void);
/* Measurement aid: arms the calling thread so that the NEXT launch of one of the HBM-bound kernels (fused
 * BN+act, hd_yolo_loss_bwd_dense, hd_roi_align_fwd/_bwd fast paths, hd_pairwise_iou) records the kernel's own
 * begin / end into the two hipEvent_t (timing enabled) — the interval a profiler reports for the dispatch, free of
 * host launch gaps.  NULL, NULL disarms.  Results are unaffected. */
hd_status hd_profile_time_next_launch(void* start_event, void* stop_event);

/* ---- anchors (FasterRCNN RPN, README.md:8) --------------------------------- */

/* HOST helper. Base anchors for one level: for each ratio r (=h/w) then each size z:
 *   hr = sqrtf(r); w = z / hr; h = z * hr; anchor = (-w/2, -h/2, w/2, h/2).
 * out_base: host [n_ratios*n_sizes, 4].  Pure host arithmetic, no GPU. */
hd_status hd_anchor_base(const float* sizes, int n_sizes, const float* ratios, int n_ratios,
                         float* out_base);

/* Tiles per-level base anchors over the feature grids.
 * Order of the output is level-major, then y, x, anchor:
 *   out[off_l + ((y*W_l + x)*A + a)] = base_l[a] + ((x+offset)*stride_l, (y+offset)*stride_l) x2
 * feat_h, feat_w, strides: host [n_levels]; base_anchors: host [n_levels, A, 4];
 * out: device [sum_l H_l*W_l*A, 4], 16-byte aligned. */
hd_status hd_anchor_grid(int n_levels, const int* feat_h, const int* feat_w, const float* strides,
                         const float* base_anchors, int A, float offset, float* out,
                         hd_stream stream);

/* ---- pairwise IoU (assigners; SURVEY.md §8a "Batched pairwise IoU") -------- */

/* Materialised IoU.  a: [B,N,4] with element stride a_bstride between images (N*4 when dense,
 * 0 to share one set of boxes, e.g. the anchors, across the batch), b: [B,M,4],
 * m_valid: [B] int32 or NULL (all M valid), out: [B,N,M].
 * Columns m >= m_valid[b] are written as -1.
 * iou = inter / (area_a + area_b - inter), 0 when the union is <= 0.
 * a, b must be 16-byte aligned. */
hd_status hd_pairwise_iou(const float* a, int64_t a_bstride, const float* b,
                          const int32_t* m_valid, int B, int N, int M, float* out,
                          hd_stream stream);

/* Fused IoU + row max/argmax (per box in a) + column max/argmax (per box in b),
 * without materialising [B,N,M].
 *   max_iou[B,N], argmax[B,N]  : best b for each a  (ties -> lowest m); -1/-1 if no valid b.
 *   gt_max_iou[B,M], gt_argmax[B,M]: best a for each b (ties -> lowest n); -1/-1 for padded b.
 * gt_* may both be NULL to skip the column reduction (then workspace may be NULL). */
size_t hd_iou_assign_workspace_bytes(int B, int N, int M);
hd_status hd_iou_assign(const float* a, int64_t a_bstride, const float* b,
                        const int32_t* m_valid, int B, int N, int M, float* max_iou,
                        int32_t* argmax, float* gt_max_iou, int32_t* gt_argmax, void* workspace,
                        size_t workspace_bytes, hd_stream stream);

/* ---- box delta codec (RPN / RoI head regression targets) ------------------- */

/* deltas = encode(src -> dst): dx=(cx_d-cx_s)/w_s, dy, dw=log(w_d/w_s), dh. src,dst,out: [n,4]. */
hd_status hd_box_encode(const float* src, const float* dst, int64_t n, float* out_deltas,
                        hd_stream stream);
/* boxes = decode(anchors, deltas) with dw,dh clamped to <= log(1000/16), then clipped to
 * [0,img_w] x [0,img_h] when img_w > 0.  anchors, deltas, out: [n,4]. */
hd_status hd_box_decode(const float* anchors, const float* deltas, int64_t n, float img_h,
                        float img_w, float* out_boxes, hd_stream stream);

/* ---- NMS (RPN class-agnostic / YOLOv5 per-class) ------------ */

/* Batched greedy NMS, exact (bit-identical to the serial algorithm):
 *   candidates of image b = { i < n_valid[b] : scores[b,i] > score_thr } (NaN never passes);
 *   order = score descending, ties -> lower index i; only the first `max_candidates`
 *   of that order are considered (<=0: all);
 *   box j is suppressed by an earlier kept box i iff (cls==NULL || cls[i]==cls[j]) and
 *   iou(i,j) > iou_thr (strict);
 *   keep[b, 0:count] = kept indices i in that order, stops at max_out; tail = -1.
 * boxes [B,N,4] (16-B aligned), scores [B,N], cls [B,N] int32 or NULL, n_valid [B] or NULL. */
size_t hd_nms_workspace_bytes(int B, int N, int max_candidates);
hd_status hd_nms_batched(const float* boxes, const float* scores, const int32_t* cls,
                         const int32_t* n_valid, int B, int N, float score_thr, float iou_thr,
                         int max_candidates, int max_out, int32_t* keep, int32_t* keep_count,
                         void* workspace, size_t workspace_bytes, hd_stream stream);

/* ---- RPN proposals (README.md:8; SURVEY.md §8a "RPN proposal top-K + NMS") -- */

/* Per image: per level top-`pre_nms_topk` by objectness logit (ties -> lower index),
 * decode deltas against anchors, clip to the image, drop boxes with w or h < min_size,
 * NMS per level (level acts as the class id), keep the best `post_nms_topk` overall
 * by logit.  Outputs are padded with zeros; scores are sigmoid(logit).
 * logits [B,NA], deltas [B,NA,4], anchors [NA,4] in the hd_anchor_grid order,
 * level_sizes: host [n_levels], sum = NA.  rois [B,post,4], roi_scores [B,post], count [B].
 * roi_index [B,post] int32 or NULL: the anchor index (0..NA-1) each kept proposal came from, in output
 * order, -1 padded — the NMS keep set itself, for callers (and parity tests) that need indices
 * rather than boxes. */
size_t hd_rpn_proposals_workspace_bytes(int B, int NA, int n_levels, int pre_nms_topk,
                                        int post_nms_topk);
hd_status hd_rpn_proposals(const float* logits, const float* deltas, const float* anchors, int B,
                           int NA, int n_levels, const int* level_sizes, int pre_nms_topk,
                           int post_nms_topk, float iou_thr, float min_size, float img_h,
                           float img_w, float* rois, float* roi_scores, int32_t* roi_index,
                           int32_t* count, void* workspace, size_t workspace_bytes, hd_stream stream);

/* ---- RoIAlign over pyramid levels (README.md:65,73) ------------------------ */

typedef struct hd_roi_levels {
  int n_levels;                       /* 1..HD_MAX_LEVELS                                  */
  const void* feat[HD_MAX_LEVELS];    /* device base pointer of each level's feature map   */
  int h[HD_MAX_LEVELS], w[HD_MAX_LEVELS];
  float spatial_scale[HD_MAX_LEVELS]; /* 1/stride                                          */
  int64_t sb[HD_MAX_LEVELS], sc[HD_MAX_LEVELS], sh[HD_MAX_LEVELS], sw[HD_MAX_LEVELS];
                                      /* element strides of (batch, channel, y, x)          */
  int k_min;                          /* pyramid level number of feat[0] (e.g. 2 for P2)    */
  float canonical_scale;              /* 224                                               */
  int canonical_level;                /* 4                                                 */
  int dtype;                          /* element type of feat[]: 0 = float32, 1 = bfloat16
                                         (RoIAlign forward only; gradient maps and RoIPool
                                         are float32)                                      */
} hd_roi_levels;

/* Level of a RoI (exact form of floor(k0 + log2(sqrt(area)/s0)), clamped):
 *   lvl = #{k in (k_min, k_max] : area >= (s0 * 2^(k-k0))^2}   (index into feat[]).
 * rois [R,5] = (batch_index, x1, y1, x2, y2); out_level [R] int32. */
hd_status hd_roi_level_map(const float* rois, int R, int k_min, int n_levels,
                           float canonical_scale, int canonical_level, int32_t* out_level,
                           hd_stream stream);

/* Forward. out[r,c,ph,pw] with element strides (out_sr,out_sc,out_sh,out_sw); out_dtype 0 = float32,
 * 1 = bfloat16.  Features float32 or bfloat16 (levels->dtype), accumulation always float32.
 * Bilinear sampling as the public RoIAlign definition: bin grid sampling_ratio^2 points
 * (<=0: adaptive ceil(roi_size/P)); aligned!=0 shifts by -0.5 px and allows sub-pixel RoIs,
 * aligned==0 forces roi size >= 1; samples outside [-1, size] contribute 0.
 * Fast path: channels-last maps and output (sc == 1, out_sc == 1, 16-byte aligned rows),
 * sampling_ratio 1 or 2, P <= 16; anything else runs the any-stride kernel. */
hd_status hd_roi_align_fwd(const hd_roi_levels* levels, const float* rois, int R, int C, int P,
                           int sampling_ratio, int aligned, void* out, int out_dtype, int64_t out_sr,
                           int64_t out_sc, int64_t out_sh, int64_t out_sw, hd_stream stream);

/* Backward: scatter-adds grad_out (grad_dtype 0 = float32, 1 = bfloat16) into the float32 gradient
 * maps grad_levels->feat[] (dtype must be 0), which the caller has zeroed.  Uses float atomics:
 * summation order is not fixed. */
hd_status hd_roi_align_bwd(const hd_roi_levels* grad_levels, const float* rois, int R, int C,
                           int P, int sampling_ratio, int aligned, const void* grad_out,
                           int grad_dtype, int64_t out_sr, int64_t out_sc, int64_t out_sh,
                           int64_t out_sw, hd_stream stream);

/* Backward as a gather (csrc/roi_align.hip, roi_align_bwd_gather_kernel): WRITES every cell of every level map of
 * grad_levels (float32 or bfloat16 maps: grad_levels->dtype; nothing has to be zeroed) exactly once, without atomics, in a
 * fixed summation order (RoI order): bit-reproducible.  `batch`: images per map (RoIs with other batch indices are ignored).
 * Needs channels-last maps (sc == 1, 16-byte aligned rows) and gradient (out_sc == 1), C a multiple of 64, P == 7,
 * sampling_ratio 1 or 2; workspace: one 32-byte record per RoI + the per-image ranges of the list (hd_roi_align_bwd_gather_workspace_bytes; a
 * list ordered by image lets every tile scan its own image's RoIs only; any order is accepted): hd_roi_align_bwd_gather_supported() says whether a call qualifies (else: hd_roi_align_bwd). */
int hd_roi_align_bwd_gather_supported(const hd_roi_levels* grad_levels, int R, int C, int P, int sampling_ratio,
                                      int grad_dtype, int64_t out_sc);
size_t hd_roi_align_bwd_gather_workspace_bytes(int R, int batch);
hd_status hd_roi_align_bwd_gather(const hd_roi_levels* grad_levels, int batch, const float* rois, int R, int C, int P,
                                  int sampling_ratio, int aligned, const void* grad_out, int grad_dtype, int64_t out_sr,
                                  int64_t out_sc, int64_t out_sh, int64_t out_sw, void* workspace, size_t workspace_bytes,
                                  hd_stream stream);

/* ---- RoI max-pooling over pyramid levels (README.md:73 "FasterRCNN-FPNP2-RoIPooling") ---- */

/* Quantised RoI max-pooling (the public RoIPool definition): the RoI corners are rounded to
 * cells of its level (round half away from zero, x*spatial_scale), width/height = end-start+1
 * (>= 1), bin (ph,pw) covers rows [floor(ph*bh), ceil((ph+1)*bh)) + start, clipped to the map;
 * out = max over the bin (first maximum in row-major order wins; an empty bin gives 0),
 * argmax = y*W + x of the winner inside its level's map, or -1 for an empty bin.
 * out and argmax [R,C,P,P] share the element strides (out_sr,out_sc,out_sh,out_sw).
 * The level of a RoI is hd_roi_level_map's. */
hd_status hd_roi_pool_fwd(const hd_roi_levels* levels, const float* rois, int R, int C, int P,
                          float* out, int32_t* argmax, int64_t out_sr, int64_t out_sc,
                          int64_t out_sh, int64_t out_sw, hd_stream stream);

/* Backward: grad_feat[level][b, c, argmax] += grad_out (float atomics; the caller zeroes the
 * grad buffers given in grad_levels->feat[]). */
hd_status hd_roi_pool_bwd(const hd_roi_levels* grad_levels, const float* rois, int R, int C,
                          int P, const float* grad_out, const int32_t* argmax, int64_t out_sr,
                          int64_t out_sc, int64_t out_sh, int64_t out_sw, hd_stream stream);

/* ---- YOLOv5 head ops (README.md:9) ----------------------------------------- */

typedef struct hd_yolo_levels {
  int n_levels;                      /* 1..HD_MAX_LEVELS                                    */
  int na;                            /* anchors per cell (<= HD_MAX_ANCHORS)                */
  int no;                            /* 5 + num_classes                                     */
  float* pred[HD_MAX_LEVELS];        /* raw head output of each level (or its grad buffer)  */
  int h[HD_MAX_LEVELS], w[HD_MAX_LEVELS];
  float stride[HD_MAX_LEVELS];
  float anchor_w[HD_MAX_LEVELS][HD_MAX_ANCHORS]; /* pixels */
  float anchor_h[HD_MAX_LEVELS][HD_MAX_ANCHORS];
  /* element strides of pred viewed as [B, na, H, W, no] */
  int64_t sb[HD_MAX_LEVELS], sa[HD_MAX_LEVELS], sy[HD_MAX_LEVELS], sx[HD_MAX_LEVELS],
      so[HD_MAX_LEVELS];
} hd_yolo_levels;

/* Grid decode (v6.1 form): xy=(2*sig(t)-0.5+grid)*stride, wh=(2*sig(t))^2*anchor, sig(rest).
 * out [B, n_total, no] = (cx, cy, w, h, obj, cls...), rows ordered level, anchor, y, x. */
hd_status hd_yolo_decode(const hd_yolo_levels* lv, int B, float* out, hd_stream stream);

/* Target assignment (anchor-shape matching + 2 neighbour cells, v6.1 build_targets form).
 * gt_boxes [B,M,4] xyxy pixels, gt_cls [B,M] int32, m_valid [B].
 * Fixed slot layout per level: slot = ((k*na + a)*B + b)*M + t, k in 0..4
 * (0 centre, 1 left, 2 up, 3 right, 4 down).
 *   slot_cell [n_levels, S] int32 = ((b*na + a)*H + gj)*W + gi, or -1 when unmatched
 *   slot_tbox [n_levels, S, 4]    = (gx - gi, gy - gj, gw, gh) in grid units
 *   slot_cls  [n_levels, S] int32
 * S = 5*na*B*M. */
hd_status hd_yolo_assign(const hd_yolo_levels* lv, const float* gt_boxes, const int32_t* gt_cls,
                         const int32_t* m_valid, int B, int M, float anchor_t,
                         int32_t* slot_cell, float* slot_tbox, int32_t* slot_cls,
                         hd_stream stream);

typedef struct hd_yolo_loss_cfg {
  float box_gain, obj_gain, cls_gain; /* hyp.box, hyp.obj, hyp.cls (already rescaled)        */
  float balance[HD_MAX_LEVELS];       /* objectness weight per level                         */
  /* Focal variant of the BCE terms ("YOLOv5s-focalloss" rows, README.md:118-154), the public
   * FocalLoss-over-BCEWithLogits form:  bce * (1 - p_t)^gamma * (z*alpha + (1-z)*(1-alpha)),
   * p_t = z*p + (1-z)*(1-p), p = sigmoid(x), z the (possibly soft) target. */
  float focal_gamma;                  /* <= 0: plain BCE everywhere                          */
  float focal_alpha;                  /* <= 0: no class-balance factor                       */
  int focal_terms;                    /* bit 0: classification term, bit 1: objectness term  */
} hd_yolo_loss_cfg;

/* Loss forward.  tobj [sum_l B*na*H_l*W_l] scratch (objectness targets, level-major),
 * acc [n_levels*4] float scratch (sum(1-ciou), n_pos, sum bce_cls, sum bce_obj per level),
 * loss_out [4] = (total, lbox, lobj, lcls) with
 *   lbox = box_gain * sum_l mean(1-ciou); lcls = cls_gain * sum_l mean BCE(cls);
 *   lobj = obj_gain * sum_l balance_l * mean BCE(obj, tobj); total = B*(lbox+lobj+lcls);
 *   tobj[cell] = max over slots on that cell of clamp(ciou, 0)  (duplicates -> max). */
size_t hd_yolo_loss_workspace_bytes(const hd_yolo_levels* lv, int B);
hd_status hd_yolo_loss_fwd(const hd_yolo_levels* lv, int B, int M, const int32_t* slot_cell,
                           const float* slot_tbox, const int32_t* slot_cls,
                           const hd_yolo_loss_cfg* cfg, float* loss_out, void* workspace,
                           size_t workspace_bytes, hd_stream stream);
/* Loss backward: writes d(total)/d(pred) * grad_scale[0] densely into grad_lv->pred[l]
 * (any strides; dense channels-last is the fast path).  workspace must be the one
 * hd_yolo_loss_fwd filled.  grad_scale: device [1] float (upstream gradient of `total`).
 * grad_bias: NULL, or device [n_levels, na*no], ZEROED BY THE CALLER, into which the
 * sum-over-cells of the gradient per head channel is accumulated (= the gradient of the Detect
 * conv's bias; computed here so the framework need not reduce the dense gradient). */
hd_status hd_yolo_loss_bwd(const hd_yolo_levels* lv, const hd_yolo_levels* grad_lv, int B, int M,
                           const int32_t* slot_cell, const float* slot_tbox,
                           const int32_t* slot_cls, const hd_yolo_loss_cfg* cfg,
                           const float* grad_scale, float* grad_bias, const void* workspace,
                           size_t workspace_bytes, hd_stream stream);

/* The two stages of hd_yolo_loss_bwd as separate launches (same arguments): _dense writes the
 * whole gradient tensor (zeros + objectness term; the HBM-bound 4*B*n*no-byte write) and leaves
 * per-workgroup objectness bias partials in the workspace; _pos (must follow _dense on the same
 * stream, same grad_lv) adds the box/class terms of the matched cells with float atomics and
 * folds everything into grad_bias. */
hd_status hd_yolo_loss_bwd_dense(const hd_yolo_levels* lv, const hd_yolo_levels* grad_lv, int B,
                                 const hd_yolo_loss_cfg* cfg, const float* grad_scale,
                                 float* grad_bias, const void* workspace, size_t workspace_bytes,
                                 hd_stream stream);
hd_status hd_yolo_loss_bwd_pos(const hd_yolo_levels* lv, const hd_yolo_levels* grad_lv, int B, int M,
                               const int32_t* slot_cell, const float* slot_tbox,
                               const int32_t* slot_cls, const hd_yolo_loss_cfg* cfg,
                               const float* grad_scale, float* grad_bias, const void* workspace,
                               size_t workspace_bytes, hd_stream stream);

/* The same two stages for a BFLOAT16 gradient: grad_lv->pred[l] points at bfloat16 elements (behind the declared float
 * pointer) with grad_lv's element strides, which must describe padded pixel rows — so == 1, sa == no, sx = pitch with
 * pitch % 8 == 0, rows contiguous, 16-byte aligned — and lv must be dense channels-last (plain or padded): the layout of
 * hd_head_conv_fwd's output, whose backward (hd_head_conv_bwd) takes exactly these rows, padding columns zero.  The
 * objectness term is rounded once; the box / class terms of a matched cell are added with the packed bfloat16 atomic add
 * (global_atomic_pk_add_bf16) on the element's 32-bit word, every addition rounded (cells matched by several targets
 * accumulate in bfloat16, as a half-precision index_put_(accumulate) would).  grad_bias sums the unrounded terms in float32 as before. */
hd_status hd_yolo_loss_bwd_dense_bf16(const hd_yolo_levels* lv, const hd_yolo_levels* grad_lv, int B,
                                      const hd_yolo_loss_cfg* cfg, const float* grad_scale, float* grad_bias,
                                      const void* workspace, size_t workspace_bytes, hd_stream stream);
hd_status hd_yolo_loss_bwd_pos_bf16(const hd_yolo_levels* lv, const hd_yolo_levels* grad_lv, int B, int M,
                                    const int32_t* slot_cell, const float* slot_tbox, const int32_t* slot_cls,
                                    const hd_yolo_loss_cfg* cfg, const float* grad_scale, float* grad_bias,
                                    const void* workspace, size_t workspace_bytes, hd_stream stream);

/* ---- CIoU on matched pairs (SURVEY.md §8a "CIoU loss fwd + bwd") ----------- */

/* pred, target [P,4] xyxy; ciou [P]; loss = 1 - ciou is left to the caller.
 * eps = 1e-7 placed as in the public v6.1 bbox_iou: h += eps, union += eps, c2 += eps;
 * alpha = v / (v - iou + 1 + eps) is treated as a constant in backward. */
hd_status hd_ciou_fwd(const float* pred, const float* target, int64_t P, float* ciou,
                      hd_stream stream);
/* grad_pred [P,4] = grad_ciou[p] * d ciou / d pred. */
hd_status hd_ciou_bwd(const float* pred, const float* target, const float* grad_ciou, int64_t P,
                      float* grad_pred, hd_stream stream);

/* ---- YOLOv5 post-processing ------------------------------------------------- */

/* From decoded predictions dec [B,n,no]: candidates with obj > conf_thr and
 * conf = obj*max_cls > conf_thr (best class only), per-class NMS (iou_thr), best `max_det`.
 * det [B,max_det,6] = (x1,y1,x2,y2,conf,cls), zero padded; det_count [B].
 * Candidate order = conf descending, ties -> lower row index; at most max_nms candidates. */
size_t hd_yolo_nms_workspace_bytes(int B, int n, int max_nms, int max_det);
hd_status hd_yolo_nms(const float* dec, int B, int n, int no, float conf_thr, float iou_thr,
                      int max_nms, int max_det, float* det, int32_t* det_count, void* workspace,
                      size_t workspace_bytes, hd_stream stream);

/* hd_yolo_decode followed by hd_yolo_nms, without materialising the decoded tensor: the candidate pass reads
 * the raw head outputs, gates on sigmoid(objectness) > conf_thr and decodes only the rows that pass.
 * Detections are identical (values, order, ties) to the two-step path. */
size_t hd_yolo_detect_workspace_bytes(const hd_yolo_levels* lv, int B, int max_nms, int max_det);
hd_status hd_yolo_detect(const hd_yolo_levels* lv, int B, float conf_thr, float iou_thr, int max_nms,
                         int max_det, float* det, int32_t* det_count, void* workspace,
                         size_t workspace_bytes, hd_stream stream);

/* ---- Weighted Boxes Fusion (README.md:19) ---------------------------------- */

/* Per image: boxes [B,K,4], scores [B,K], labels [B,K] int32, model_weight [B,K] (weight of the
 * TTA view each box came from), k_valid [B].  Boxes with score < skip_thr are dropped; per
 * label, boxes are visited by weighted score descending (ties -> lower index) and join the
 * cluster whose fused box has the highest IoU > iou_thr (ties -> older cluster), else open one;
 * the fused box is the confidence-weighted mean (conf = score*model_weight), fused score =
 * sum(conf)/n * min(n, weight_sum)/weight_sum with weight_sum = sum of the view weights
 * (= number of views when all weights are 1; "avg" confidence).  out [B,K,6] = (x1,y1,x2,y2,score,label) sorted by
 * score descending (ties -> lower label, then older cluster), zero padded; out_count [B].
 * Limits: 0 <= label < 4096, K <= 2^20. */
size_t hd_wbf_workspace_bytes(int B, int K);
hd_status hd_wbf(const float* boxes, const float* scores, const int32_t* labels,
                 const float* model_weight, const int32_t* k_valid, int B, int K, float weight_sum,
                 float iou_thr, float skip_thr, float* out, int32_t* out_count, void* workspace,
                 size_t workspace_bytes, hd_stream stream);

/* ---- backbone glue on the path: SPPF max-pool (README.md:9, YOLOv5 backbone) ---------------- */

/* 5x5 / stride 1 / pad 2 max-pool of a dense channels-last tensor [B,H,W,C] (dtype 0 = float32,
 * C % 4 == 0; 1 = bfloat16, C % 8 == 0).  idx [B,H,W,C] uint8 receives the winning tap
 * (ty*5+tx; first tap in scan order wins ties).  Backward is the gather form over idx (no
 * atomics, deterministic): grad_in[p] = sum of grad_out[q] over outputs q whose tap points at p. */
/* in_row_stride / out_row_stride / grad_out_row_stride: elements between consecutive pixels (0 or C = dense;
 * > C, multiple of the vector width = the tensor is a channel slice of a wider channels-last tensor, e.g. a slot of
 * the concatenation buffer SPPF's four branches are written into).  idx and grad_in are always dense. */
hd_status hd_maxpool5_fwd(const void* in, int64_t in_row_stride, int dtype, int B, int H, int W, int C,
                          void* out, int64_t out_row_stride, uint8_t* idx, hd_stream stream);
hd_status hd_maxpool5_bwd(const void* grad_out, int64_t grad_out_row_stride, const uint8_t* idx, int dtype,
                          int B, int H, int W, int C, void* grad_in, hd_stream stream);
/* 3x3 / stride 2 / pad 1 max-pool (the ResNet50 stem, reference README.md:8) on DENSE channels-last tensors: in [B, H, W, C]
 * -> out [B, Ho, Wo, C] with Ho = (H - 1) / 2 + 1; idx: one byte per OUTPUT element (winning tap 0..8), dense.  Backward is
 * the gather form (an input pixel sits under at most 2 x 2 windows): grad_in [B, H, W, C] from grad_out / idx, no atomics. */
hd_status hd_maxpool3s2_fwd(const void* in, int dtype, int B, int H, int W, int C, void* out, uint8_t* idx, hd_stream stream);
hd_status hd_maxpool3s2_bwd(const void* grad_out, const uint8_t* idx, int dtype, int B, int H, int W, int C, void* grad_in,
                            hd_stream stream);

/* Fused BatchNorm2d (batch statistics) + activation on a dense channels-last tensor seen as rows
 * [R = B*H*W, C]; dtype 0 = float32 (C % 4 == 0), 1 = bfloat16 (C % 8 == 0); C/vec <= 256.
 * act: 0 none, 1 SiLU, 2 ReLU.  gamma/beta/running_* / save_* are float32 [C].
 * Accumulators are replicated: sums / dsums are [hd_bn_act_replicas(), 2, C] float32.
 *   forward : _stats accumulates per-channel (sum, sum of squares) into sums (ZEROED BY THE
 *             CALLER); _apply then writes y = act(gamma*(x-mean)*invstd + beta), saves mean/invstd
 *             and updates running_mean/var in place when they are non-NULL (momentum, unbiased var).
 *   infer   : _infer uses the running statistics.
 *   backward: _bwd_reduce accumulates (sum dz, sum dz*xhat) into dsums (ZEROED BY THE CALLER),
 *             dz = dy*act'(z) with z recomputed from x; _bwd_dx writes dx and dgamma/dbeta. */
int hd_bn_act_replicas(void);
/* Optional SECOND PLACE of a [R, C] operand of the fused BN kernels (y of _apply / _infer, dy of the backward pair):
 * channels [0, split) live at the primary pointer (row stride given beside it; 0 = dense over `split` channels),
 * channels [split, C) at p2 with row stride ld2 (0 = dense over C - split).  NULL (or p2 == NULL) = one place.
 * Serves the merged branch convolutions of a C3 block (cv1 | cv2 as ONE convolution + ONE BatchNorm): the first half
 * of the activation is a dense tensor for the bottleneck stack, the second half is written straight into the slot of
 * the block's concatenation; backward reads the two incoming gradients from where they are.  split, ld2: multiples of
 * the vector width (4 f32 / 8 bf16); p2 16-byte aligned. */
typedef struct hd_rows2 {
  const void* p2;
  int64_t ld2;
  int split;
} hd_rows2;
hd_status hd_bn_act_stats(const void* x, int dtype, int64_t R, int C, float* sums, hd_stream stream);
/* stat_rows (apply, bwd_dx): rows the (d)sums cover; 0 = R.  > R when the caller all-reduced the sums across
 * ranks between the two launches (synchronised BatchNorm over a data-parallel group). */
hd_status hd_bn_act_apply(const void* x, int dtype, int64_t R, int64_t stat_rows, int C, const float* sums,
                          const float* gamma, const float* beta, float eps, float momentum,
                          float* running_mean, float* running_var, float* save_mean,
                          float* save_invstd, int act, const void* res, int res_mode, void* y,
                          int64_t y_row_stride, const hd_rows2* y_tail, hd_stream stream);
/* y_row_stride (apply, infer): elements between consecutive rows of y; 0 (or C) = dense, > C = y is a channel slice of a
 * wider channels-last tensor — the slot of a following concatenation, so the activation is written where torch.cat
 * would otherwise copy it (multiple of the vector width). */
/* Residual input (same dtype/shape as x, dense rows): res_mode 0 = none (res NULL); 1 = y = act(bn(x) + res), the
 * ResNet bottleneck tail; 2 = y = act(bn(x)) + res, the YOLOv5 bottleneck shortcut.  Backward: only mode 1 needs the
 * kernels' help — pass the same `res` to _bwd_reduce and _bwd_dx (act' is taken at bn(x) + res) and _bwd_dx also
 * writes d(res) = dz into dres; for mode 2 d(res) is dy itself and the backward entry points are called with NULL. */
/* Bias + activation of a biased convolution that was run WITHOUT its bias (the RPN head's conv -> ReLU; the reference's
 * FasterRCNN, README.md:8): y = act(x + bias[c]) on channels-last rows [R, C] (dtype 0 float32 / 1 bfloat16, C a multiple
 * of the 16-byte vector, dense rows; y may alias x).  Backward from the OUTPUT: dx = dy * act'(y) (act 2 = ReLU: y > 0;
 * act 0: dx = dy, pass dx NULL to skip the write) and d(bias)[c] = sum over rows of dx, added into the replicated
 * accumulators dsums [hd_bn_act_replicas()][2][C] float32 (first half of every replica; zeroed by the caller). */
hd_status hd_bias_act_fwd(const void* x, int dtype, int64_t R, int C, const float* bias, int act, void* y, hd_stream stream);
hd_status hd_bias_act_bwd(const void* y, const void* dy, int dtype, int64_t R, int C, int act, void* dx, float* dsums,
                          hd_stream stream);
hd_status hd_bn_act_infer(const void* x, int dtype, int64_t R, int C, const float* gamma,
                          const float* beta, const float* running_mean, const float* running_var,
                          float eps, int act, void* y, int64_t y_row_stride, const hd_rows2* y_tail,
                          hd_stream stream);
/* dy_row_stride: elements between consecutive rows of dy; 0 (or C) = dense rows, > C = dy is a channel slice
 * of a wider channels-last tensor (the gradient of a concatenation input), multiple of the vector width. */
hd_status hd_bn_act_bwd_reduce(const void* x, const void* dy, int64_t dy_row_stride, int dtype, int64_t R, int C,
                               const float* save_mean, const float* save_invstd, const float* gamma,
                               const float* beta, int act, const void* res, float* dsums, const hd_rows2* dy_tail,
                               hd_stream stream);
hd_status hd_bn_act_bwd_dx(const void* x, const void* dy, int64_t dy_row_stride, int dtype, int64_t R,
                           int64_t stat_rows, int C, const float* save_mean, const float* save_invstd, const float* gamma,
                           const float* beta, int act, const void* res, const float* dsums, void* dx,
                           void* dres, float* dgamma, float* dbeta, const hd_rows2* dy_tail, hd_stream stream);

/* ---- 1x1 convolution on channels-last rows, BatchNorm statistics in the epilogue (csrc/pointwise_conv.hip) ----
 * The convolution half of the reference's Conv+BN+SiLU block (README.md:9 "YOLOv5 ... backbone/neck"; the block
 * itself is NOT IN SNAPSHOT, SURVEY.md section 8a "Backbone + neck fwd/bwd") for kernel 1x1, stride 1, groups 1, no
 * bias — 34 of YOLOv5s' 57 convolutions:
 *     y[R, Cout] = x[R, Cin] . w[Cout, Cin]^T      bfloat16 in / out, float32 accumulate (MFMA)
 * x rows may be a channel slice of a wider channels-last tensor (x_row_stride elements between rows, 0 = Cin, multiple
 * of 8); likewise y (y_row_stride, 0 = Cout).  w is dense [Cout, Cin].  Cin, Cout multiples of 32, Cout <= 1024.
 * stats (nullable): float32 [hd_bn_act_replicas(), 2, Cout], ZEROED BY THE CALLER — per-channel (sum, sum of squares)
 * of the bf16-rounded y are ADDED, exactly what hd_bn_act_stats(y) would add, so hd_bn_act_apply can follow directly.
 * With transposed weights wt[Cin, Cout] the same entry point is the data gradient (any supported shape):
 *     dx[R, Cin] = dy[R, Cout] . wt^T   ->  hd_pointwise_conv(dy, ld, wt, R, Cout, Cin, dx, ld, NULL, s). */
int hd_pointwise_conv_supported(int64_t R, int Cin, int Cout);
/* Data gradient straight from the forward weight w[Cout, Cin] (its columns are read through LDS transpose reads, no
 * transposed copy): dx[R, Cin] = dy[R, Cout] . w.  Cout in {32, 64, 128, 256} (the weight slab must stay in LDS);
 * other shapes: HD_ERR_SHAPE, and the caller uses hd_pointwise_conv with a transposed copy or the library. */
int hd_pointwise_conv_resident(int64_t R, int Cin, int Cout);  /* 1: streaming form (weights resident in LDS) */
int hd_pointwise_conv_dgrad_supported(int64_t R, int Cin, int Cout);
hd_status hd_pointwise_conv_dgrad(const void* dy, int64_t dy_row_stride, const void* w, int64_t R, int Cin, int Cout, void* dx,
                                  int64_t dx_row_stride, hd_stream stream);
hd_status hd_pointwise_conv(const void* x, int64_t x_row_stride, const void* w, int64_t R, int Cin, int Cout, void* y,
                            int64_t y_row_stride, float* stats, hd_stream stream);

/* The row-sliced float32 partial products of a weight gradient that still have to be summed into dw (bfloat16, `count`
 * elements, a multiple of 4): hd_*_wgrad_partial fills one; it is consumed either by hd_wgrad_reduce (a launch of its
 * own) or as the `piggyback` argument of the NEXT hd_*_wgrad_partial call on the same stream, whose launch then carries
 * the summation in extra workgroups — one launch per layer instead of two.  The slab must stay untouched until then. */
typedef struct hd_wgrad_job {
  const float* slab;
  int slices;
  int64_t count;
  void* dw;
} hd_wgrad_job;
hd_status hd_wgrad_reduce(const hd_wgrad_job* job, hd_stream stream);

/* hd_pointwise_conv_dgrad whose result is ADDED to `add` (bfloat16 rows [R, Cin], add_row_stride elements, 0 = dense)
 * before it is stored: dx = dy . w + add, the two bfloat16 values summed in float32 and rounded once — the gradient of a
 * layer input that also feeds a shortcut, without the separate add launch.  dx may alias add. */
hd_status hd_pointwise_conv_dgrad_add(const void* dy, int64_t dy_row_stride, const void* w, int64_t R, int Cin, int Cout,
                                      const void* add, int64_t add_row_stride, void* dx, int64_t dx_row_stride,
                                      hd_stream stream);
/* hd_pointwise_wgrad split in its two halves: this call launches the partial products (piggyback nullable) and describes the
 * pending summation in *job_out (dw is written by whoever runs the job). */
hd_status hd_pointwise_wgrad_partial(const void* dy, int64_t dy_row_stride, const void* x, int64_t x_row_stride, int64_t R,
                                     int Cin, int Cout, void* dw, void* workspace, size_t workspace_bytes,
                                     const hd_wgrad_job* piggyback, hd_wgrad_job* job_out, hd_stream stream);

/* The WHOLE backward of a 1x1 convolution in one launch (csrc/pointwise_bwd.hip; README.md:9 — every C3 / Bottleneck block
 * of the reference's CSPDarknet / PAFPN is 1x1-heavy): dx[R, Cin] = dy[R, Cout] . w (+ add, as hd_pointwise_conv_dgrad_add;
 * add nullable) AND the row-sliced float32 partial products of dw[Cout, Cin] = dy^T . x, from ONE staging of dy.  Replaces
 * the pair hd_pointwise_conv_dgrad[_add] + hd_pointwise_wgrad_partial where Cout is 32 / 64 / 128 / 256 (the reduction of
 * the data gradient) and Cin a multiple of 32 with at most 32 column tiles.  w: the forward weight [Cout][Cin] as it lies.
 * The slab holds ONE partial per workgroup of a column tile (hd_pointwise_bwd_workspace_bytes: 4-8 MB whatever the layer);
 * piggyback / job_out as for hd_pointwise_wgrad_partial.  dx may alias add. */
int hd_pointwise_bwd_supported(int64_t R, int Cin, int Cout);
size_t hd_pointwise_bwd_workspace_bytes(int64_t R, int Cin, int Cout);
hd_status hd_pointwise_bwd(const void* dy, int64_t dy_row_stride, const void* w, const void* x, int64_t x_row_stride, int64_t R,
                           int Cin, int Cout, const void* add, int64_t add_row_stride, void* dx, int64_t dx_row_stride, void* dw,
                           void* workspace, size_t workspace_bytes, const hd_wgrad_job* piggyback, hd_wgrad_job* job_out,
                           hd_stream stream);

/* The Detect convolution of the YOLOv5 head (README.md:9: "Head: YOLOv5Head"; one biased 1x1 convolution per pyramid level
 * to Cout = na (5 + nc) channels — 255 for 80 classes, no tile multiple).  Rows of the raw head output are PADDED to
 * hd_head_conv_pitch(Cout) columns (the next of 32 / 64 / 128 / 256; 0 = Cout > 256, unsupported) so that they stay 16-byte
 * aligned and both convolution and loss kernels stream them; the loss reads the first Cout columns through the
 * [B, na, H, W, no] strided view (hd_yolo_levels with sx = pitch), the padding columns are written as zeros.
 *   fwd: y[R, pitch] (float32, y_row_stride >= pitch elements, 0 = dense) = x[R, Cin] . w[Cout, Cin]^T + bias[Cout];
 *        x, w bfloat16, bias float32; Cin 32 / 64 / 128 / 256 (the resident-weight streaming kernel of hd_pointwise_conv,
 *        float32 epilogue: accumulator + bias is stored unrounded).
 *   bwd: as hd_pointwise_bwd with dy[R, pitch] bfloat16 whose padding columns are ZERO: dx[R, Cin] = dy . w and the
 *        row-sliced partials of dw[Cout, Cin] (bfloat16 after the slice sum; job_out / piggyback as there), one launch.
 *        Cin a multiple of 32 with at most 32 column tiles.  The bias gradient comes from the loss kernels
 *        (hd_yolo_loss_bwd_*: grad_bias). */
int hd_head_conv_pitch(int Cout);
int hd_head_conv_fwd_supported(int64_t R, int Cin, int Cout);
hd_status hd_head_conv_fwd(const void* x, int64_t x_row_stride, const void* w, const float* bias, int64_t R, int Cin, int Cout,
                           float* y, int64_t y_row_stride, hd_stream stream);
int hd_head_conv_bwd_supported(int64_t R, int Cin, int Cout);
size_t hd_head_conv_bwd_workspace_bytes(int64_t R, int Cin, int Cout);
hd_status hd_head_conv_bwd(const void* dy, int64_t dy_row_stride, const void* w, const void* x, int64_t x_row_stride, int64_t R,
                           int Cin, int Cout, void* dx, int64_t dx_row_stride, void* dw, void* workspace, size_t workspace_bytes,
                           const hd_wgrad_job* piggyback, hd_wgrad_job* job_out, hd_stream stream);

/* The YOLOv5 v6.x stem, Conv2d(3, Cout, kernel 6, stride 2, padding 2, no bias), on channels-last images (csrc/stem_conv.hip;
 * the reference's backbone list, README.md:9).  x: [B, H, W, 3] float32 (x_dtype 0) or bfloat16 (1), dense, H even,
 * W % 4 == 0; w: bfloat16 [Cout][6][6][3] (the channels-last memory of a [Cout, 3, 6, 6] weight); Cout 32 or 64;
 * y: bfloat16 rows [B H/2 W/2, Cout] (y_row_stride elements, 0 = dense).  stats as for hd_pointwise_conv.
 * The weight gradient comes as row-sliced partial products + a pending hd_wgrad_job (dw: bfloat16 [Cout][6][6][3]);
 * workspace: hd_stem_conv_wgrad_workspace_bytes(...).  The images need no gradient. */
int hd_stem_conv_supported(int B, int H, int W, int Cout, int x_dtype);
size_t hd_stem_conv_wgrad_workspace_bytes(int B, int H, int W, int Cout);
hd_status hd_stem_conv_fwd(const void* x, int x_dtype, const void* w, int B, int H, int W, int Cout, void* y,
                           int64_t y_row_stride, float* stats, hd_stream stream);
hd_status hd_stem_conv_wgrad_partial(const void* dy, int64_t dy_row_stride, const void* x, int x_dtype, int B, int H, int W,
                                     int Cout, void* dw, void* workspace, size_t workspace_bytes, hd_wgrad_job* job_out,
                                     hd_stream stream);

/* Weight gradient of the same convolution: dw[Cout, Cin] (bfloat16) = dy[R, Cout]^T . x[R, Cin], float32 accumulate.
 * Two launches (row-sliced partial products into a float32 slab in `workspace`, then the sum over slices), no zero-fill.
 * Cin, Cout multiples of 64 (64- / 128-wide tiles), or multiples of 32 up to 256 (32 x 32 tiles: hd_pointwise_wgrad_supported
 * says).  workspace: hd_pointwise_wgrad_workspace_bytes(R, Cin, Cout), 16-byte aligned, contents undefined on entry and exit. */
int hd_pointwise_wgrad_supported(int64_t R, int Cin, int Cout);
size_t hd_pointwise_wgrad_workspace_bytes(int64_t R, int Cin, int Cout);
hd_status hd_pointwise_wgrad(const void* dy, int64_t dy_row_stride, const void* x, int64_t x_row_stride, int64_t R, int Cin,
                             int Cout, void* dw, void* workspace, size_t workspace_bytes, hd_stream stream);

/* 3x3 / stride 1 / pad 1 convolution forward on channels-last rows, BatchNorm statistics in the epilogue like
 * hd_pointwise_conv (the bottleneck 3x3 of the YOLOv5 C3 blocks):
 *     y[b, h, w, n] (bfloat16) = sum over (ky, kx, c) of x[b, h + ky - 1, w + kx - 1, c] * w[n][ky][kx][c]
 * w: a channels-last [Cout, Cin, 3, 3] bfloat16 weight, i.e. memory [Cout][3][3][Cin].  W in {20, 26, 40, 52, 80, 104}; Cin a
 * multiple of 32, Cout of 64 (csrc/conv3x3.hip); also W = 160 with Cin = Cout = 32 (csrc/conv3x3_c32.hip); row strides in
 * elements (0 = dense, multiples of 8).  stats: nullable, as for hd_pointwise_conv.
 * The data gradient is the same entry point applied to dy with the weights flipped and transposed:
 *     wt[c][ky][kx][n] = w[n][2 - ky][2 - kx][c]   ->   dx = hd_conv3x3_fwd(dy, wt, ..., Cin = Cout, Cout = Cin). */
int hd_conv3x3_fwd_supported(int B, int H, int W, int Cin, int Cout);
hd_status hd_conv3x3_fwd(const void* x, int64_t x_row_stride, const void* w, int B, int H, int W, int Cin, int Cout, void* y,
                         int64_t y_row_stride, float* stats, hd_stream stream);

/* Forward of the first stride-2 3x3 convolution of the YOLOv5 backbone (reference README.md:9: the CSPDarknet backbone's second
 * Conv+BN+SiLU block; csrc/conv3x3s2_fwd.hip): 3x3 / stride 2 / pad 1, Cin = 32,
 * Cout = 64, Wout = 160, the input exactly 2 Hout x 2 Wout; BatchNorm statistics in the epilogue like hd_pointwise_conv:
 *     y[b, i, j, n] (bfloat16) = sum over (ky, kx, c) of x[b, 2 i - 1 + ky, 2 j - 1 + kx, c] * w[n][ky][kx][c]
 * w: channels-last [Cout, Cin, 3, 3] bfloat16 (memory [Cout][3][3][Cin]); stats: nullable, as for hd_pointwise_conv. */
int hd_conv3x3s2_fwd_supported(int B, int Hout, int Wout, int Cin, int Cout);
hd_status hd_conv3x3s2_fwd(const void* x, int64_t x_row_stride, const void* w, int B, int Hout, int Wout, int Cin, int Cout,
                           void* y, int64_t y_row_stride, float* stats, hd_stream stream);

/* Weight gradient of the same layer (README.md:9; csrc/conv3x3s2_wgrad.hip; hd_conv3x3_wgrad's kernels need 64-channel tiles and do not take it):
 *     dw[n][ky][kx][c] (bfloat16, channels-last memory of [Cout, Cin, 3, 3]) = sum over (b, i, j) of
 *         dy[b, i, j, n] * x[b, 2 i - 1 + ky, 2 j - 1 + kx, c]
 * Two launches: per-workgroup float32 partials into `workspace`, then their sum (no zero-fill, no atomics).
 * workspace: hd_conv3x3s2_wgrad_c32_workspace_bytes(...), 16-byte aligned, contents undefined on entry and exit. */
int hd_conv3x3s2_wgrad_c32_supported(int B, int Hout, int Wout, int Cin, int Cout);
size_t hd_conv3x3s2_wgrad_c32_workspace_bytes(int B, int Hout, int Wout, int Cin, int Cout);
hd_status hd_conv3x3s2_wgrad_c32(const void* dy, int64_t dy_row_stride, const void* x, int64_t x_row_stride, int B, int Hout, int Wout,
                                 int Cin, int Cout, void* dw, void* workspace, size_t workspace_bytes, hd_stream stream);

/* Data gradient of the first stride-2 3x3 convolution of the YOLOv5 backbone (README.md:9; csrc/conv3x3s2_dgrad.hip): 3x3 / stride 2 / pad 1,
 * Cin = 32 (channels of dx), Cout = 64 (channels of dy), Wout = 160, the input exactly twice the output (2 Hout x 2 Wout):
 *     dx[b, p, q, c] (bfloat16) = sum over (ky, kx, n) with p = 2 i - 1 + ky, q = 2 j - 1 + kx of dy[b, i, j, n] * w[n][ky][kx][c]
 * w: the layer's own channels-last [Cout, Cin, 3, 3] bfloat16 weight (memory [Cout][3][3][Cin]; no flipped copy).  Every element
 * of dx is written exactly once (no zero-fill by the caller).  Row strides in elements (0 = dense, multiples of 8). */
int hd_conv3x3s2_dgrad_supported(int B, int Hout, int Wout, int Cin, int Cout);
hd_status hd_conv3x3s2_dgrad(const void* dy, int64_t dy_row_stride, const void* w, int B, int Hout, int Wout, int Cin, int Cout,
                             void* dx, int64_t dx_row_stride, hd_stream stream);

/* Weight gradient of a 3x3 / pad 1 convolution of stride 1 or 2 on channels-last rows (the bottleneck and the downsampling
 * 3x3 convolutions of the YOLOv5 backbone / neck):
 *     dw[n][ky][kx][k] (bfloat16, i.e. a channels-last [Cout, Cin, 3, 3] weight) = sum over (b, h, w) of
 *         dy[b, h, w, n] * x[b, stride h + ky - 1, stride w + kx - 1, k]   (zero outside the image; float32 accumulate)
 * H, W: the OUTPUT size; dy [B, H, W, Cout], x [B, stride H, stride W, Cin] bfloat16 rows (strides in elements, 0 = dense).
 * Two kernels behind one entry point: the row kernels (csrc/pointwise_conv.hip: W in {20, 26, 40, 52, 80, 104}, Cin, Cout
 * multiples of 64) and the flat-position kernel (csrc/conv3x3_wgrad.hip: any W whose row block fits LDS, Cin, Cout multiples
 * of 32); hd_conv3x3_wgrad_supported says which shapes are taken, hd_conv3x3_wgrad_select(mode) overrides the choice between
 * the two (0 automatic, 1 row kernels only, 2 the flat kernel on every shape it can run: tests and probes).
 * Two launches (row-sliced partial products for all nine taps, then the slab sum), no zero-fill.
 * workspace: hd_conv3x3_wgrad_workspace_bytes(...), 16-byte aligned, contents undefined on entry and exit. */
/* The flipped, transposed weights hd_conv3x3_fwd takes as a data gradient, wt[c][ky][kx][n] = w[n][2 - ky][2 - kx][c], for up
 * to HD_MAX_FLIP layers in ONE launch (w[l]: bfloat16 [Cout][3][3][Cin], wt[l]: bfloat16 [Cin][3][3][Cout]; the pointer and
 * size arrays are host memory).  The weights change once per optimizer step: rebuild the copies at the head of the step. */
#define HD_MAX_FLIP 32
hd_status hd_conv3x3_flip_weights(int n_layers, const void* const* w, void* const* wt, const int* cout, const int* cin,
                                  hd_stream stream);
/* The same launch for a mix of 3x3 and 1x1 layers: taps[l] = 9 (as above) or 1 (w[l]: [Cout][Cin] -> wt[l]: [Cin][Cout], a plain
 * transpose — the operand with which hd_pointwise_conv computes the DATA gradient of a 1x1 layer whose Cout is too long for the
 * transpose-fragment form of hd_pointwise_conv_dgrad: dx = hd_pointwise_conv(dy, wt, Cin = Cout, Cout = Cin)).  taps: nullable = all 9. */
hd_status hd_conv_flip_weights(int n_layers, const void* const* w, void* const* wt, const int* cout, const int* cin,
                               const int* taps, hd_stream stream);
int hd_conv3x3_wgrad_supported(int B, int H, int W, int Cin, int Cout, int stride);
void hd_conv3x3_wgrad_select(int mode);
size_t hd_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int stride);
hd_status hd_conv3x3_wgrad(const void* dy, int64_t dy_row_stride, const void* x, int64_t x_row_stride, int B, int H, int W,
                           int Cin, int Cout, int stride, void* dw, void* workspace, size_t workspace_bytes, hd_stream stream);
/* The same in its two halves (see hd_wgrad_job): partial products now, the slice sum by whoever runs *job_out; a pending
 * job of an earlier call may ride along as `piggyback` (nullable). */
hd_status hd_conv3x3_wgrad_partial(const void* dy, int64_t dy_row_stride, const void* x, int64_t x_row_stride, int B, int H,
                                   int W, int Cin, int Cout, int stride, void* dw, void* workspace, size_t workspace_bytes,
                                   const hd_wgrad_job* piggyback, hd_wgrad_job* job_out, hd_stream stream);

/* ---- nearest 2x upsampling + channel concatenation in one pass (csrc/upcat.hip) ----
 * The top-down junctions of the YOLOv5 PAFPN neck (README.md:9): out[b, y, x, :] = [small[b, y/2, x/2, :] | big[b, y, x, :]]
 * on channels-last rows.  H, W: the fine grid (even); small is [B, H/2, W/2, Cs], big [B, H, W, Cb], out [B, H, W, Cs+Cb];
 * dtype 0 float32 (channels % 4 == 0) / 1 bfloat16 (channels % 8 == 0); row strides in elements, 0 = dense.
 * Backward: d_small[b, y, x, :] = sum of the four fine gradients grad_out[b, 2y+{0,1}, 2x+{0,1}, :Cs] (float32
 * accumulate); the gradient of `big` is the channel slice grad_out[..., Cs:] itself (a view for the caller). */
hd_status hd_upsample2x_cat_fwd(const void* small, int64_t small_row_stride, const void* big, int64_t big_row_stride, int dtype,
                                int B, int H, int W, int Cs, int Cb, void* out, int64_t out_row_stride, hd_stream stream);
hd_status hd_upsample2x_cat_bwd(const void* grad_out, int64_t grad_row_stride, int dtype, int B, int H, int W, int Cs,
                                void* d_small, int64_t d_small_row_stride, hd_stream stream);

/* ---- AdamW over all tensors of a parameter group in one launch, bf16 weight shadows written in the same pass ----
 * (csrc/optim.hip; the optimizer of the train step BASELINE.json's metric times.)  `tensors` is a DEVICE array of
 * descriptors: float32 parameter / gradient / first / second moment (dense, same memory order, n elements), an optional
 * bfloat16 shadow of the parameter (NULL: none) and the tensor's weight decay.  Workgroup c handles elements
 * [chunk_start[c], chunk_start[c] + chunk_elems) of tensor chunk_tensor[c] (device arrays; chunk_elems a multiple of 1024).
 * bc1 = 1 - beta1^t, bc2 = 1 - beta2^t for the step count t (host side).  Semantics: torch.optim.AdamW. */
typedef struct hd_adamw_tensor {
  void* p;
  const void* g;
  void* m;
  void* v;
  void* shadow;
  int64_t n;
  float weight_decay;
  int32_t g_dtype; /* dtype of g: 0 = float32, 1 = bfloat16 */
} hd_adamw_tensor;
hd_status hd_adamw_multi(const hd_adamw_tensor* tensors, const int32_t* chunk_tensor, const int64_t* chunk_start, int n_chunks,
                         int chunk_elems, double lr, double beta1, double beta2, double eps, double bc1, double bc2,
                         hd_stream stream);

/* ---- Mosaic-4 augmentation on device (README.md:76-77,129-130; convention: DESIGN.md section 2 item 12) ---- */

/* images/out: float32 [B,3,S,S] with element strides (b,c,y,x); S even; out must not alias images.
 * src [B,4] int32 = source image of tile TL, TR, BL, BR (indices into the batch); center [B,2] int32 =
 * (xc, yc) on the 2S x 2S canvas, in [S/2, 3S/2]; apply [B] int32 (0: out[b] = images[src[b][0]]).
 * out(x,y) = mean of the 2x2 canvas pixels at (2x, 2y); uncovered canvas = fill. */
hd_status hd_mosaic4(const float* images, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int B, int S,
                     const int32_t* src, const int32_t* center, const int32_t* apply, float fill,
                     float* out, int64_t ob, int64_t oc, int64_t oy, int64_t ox, hd_stream stream);

/* Boxes of the same mosaic: shift, clip to the tile's visible rectangle, halve; keep iff w > min_wh and
 * h > min_wh and clipped area / original area > min_area_ratio.  gt_boxes [B,M,4] xyxy, gt_cls [B,M],
 * m_valid [B] -> out_boxes [B,4M,4], out_cls [B,4M] (zero padded), out_count [B]; order: tile, then source. */
hd_status hd_mosaic4_boxes(const float* gt_boxes, const int32_t* gt_cls, const int32_t* m_valid, int B, int M,
                           int S, const int32_t* src, const int32_t* center, const int32_t* apply,
                           float min_wh, float min_area_ratio, float* out_boxes, int32_t* out_cls,
                           int32_t* out_count, hd_stream stream);

/* ---- hipGraph hygiene ------------------------------------------------------------------------ */

/* Replaces every memset node of a captured, NOT YET INSTANTIATED hipGraph_t by an equivalent
 * fill-kernel node, and every one-span device-to-device memcpy node by a copy-kernel node, with the
 * same predecessors/successors (host call; no stream).  On ROCm 7.2 memset nodes were observed not
 * to stay ordered against neighbouring kernel nodes on replay; memcpy nodes
 * take the same engine path and are rewritten as a precaution.
 * n_nodes_out / n_replaced_out (host, may be NULL) receive the node count and the number rewritten. */
hd_status hd_graph_replace_memsets(void* hip_graph, int* n_nodes_out, int* n_replaced_out);

/* Multi-GPU overlap under hipGraph replay (reference README.md:29 DDP; north_star "all-reduce overlapped with the
 * backward pass").  hd_graph_mark: a one-lane kernel that stores `index` into *slot — launched inside a stream
 * capture right after gradient bucket `index` was packed, it marks that point of the captured graph.
 * hd_graph_split_instantiate: cuts a captured, NOT instantiated hipGraph_t into segments that end at its markers
 * (segment k = the nodes marker k depends on that no earlier marker needs; then the tail) and instantiates each
 * (host call).  out_execs [max_segments] receives the hipGraphExec_t handles in replay order, marker_index_out[k] the
 * index stored by the marker that closes segment k (only for the marker segments), n_segments_out their number
 * (markers + 1, or markers when nothing follows the last one).  The original graph must stay alive (it owns the
 * kernel parameters' memory pool on the framework side); launch the segments in order on one stream with
 * hd_graph_exec_launch and destroy them with hd_graph_exec_destroy. */
hd_status hd_graph_mark(int32_t* slot, int index, hd_stream stream);
hd_status hd_graph_split_instantiate(void* hip_graph, const int32_t* slot, int max_segments, void** out_execs,
                                     int* marker_index_out, int* n_segments_out, int* n_nodes_out);
hd_status hd_graph_exec_launch(void* exec, hd_stream stream);
hd_status hd_graph_exec_destroy(void* exec);

#ifdef __cplusplus
}
#endif
#endif /* HD_OPS_H */
