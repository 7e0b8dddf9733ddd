/*
 * hd_oracle.c — CPU restatement of the detection hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the reference snapshot (/root/reference) holds README.md, an empty
 * .gitignore and two images, and no source, tests or golden vectors.
 * There is therefore no reference file:line to follow below README level.  Every function
 * restates the PUBLISHED definition of the algorithm the README names (the README's own
 * "reference" links, README.md:158-164: bubbliiiing faster-rcnn / yolov5-v6.1, ultralytics
 * yolov5, mmdetection), under the conventions frozen in DESIGN.md §2.  A second, independent
 * restatement (oracle/torch_ref.py) cross-checks this file; fixtures in tests/golden pin both.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object.  The product (heltondetection_amd) never does.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off matters: keep masks are compared bit-for-bit with the HIP kernels, which
 * are built with the same flag and evaluate every expression in the order written here.
 *
 * All pointers are host pointers.  Plain serial loops; sizes the tests use finish in seconds.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_LEVELS 8
#define ORC_MAX_ANCHORS 16

/* ------------------------------------------------------------------------------------------
 * shared arithmetic
 * ---------------------------------------------------------------------------------------- */

/* IoU of two xyxy boxes; DESIGN.md §2.1.  (README.md:8-9: both detector families.) */
static float iou_xyxy(const float* a, const float* b) {
  float area_a = (a[2] - a[0]) * (a[3] - a[1]);
  float area_b = (b[2] - b[0]) * (b[3] - b[1]);
  float iw = fminf(a[2], b[2]) - fmaxf(a[0], b[0]);
  float ih = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
  iw = iw > 0.0f ? iw : 0.0f;
  ih = ih > 0.0f ? ih : 0.0f;
  float inter = iw * ih;
  float uni = (area_a + area_b) - inter;
  return uni > 0.0f ? inter / uni : 0.0f;
}

static float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* BCE with logits, stable form: max(x,0) - x*z + log1p(exp(-|x|)). */
static float bce_logits(float x, float z) {
  return fmaxf(x, 0.0f) - x * z + log1pf(expf(-fabsf(x)));
}

/* order-preserving map float -> uint32 (larger float -> larger key); NaN not expected. */
static uint32_t f2key(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

/* ------------------------------------------------------------------------------------------
 * anchors (FasterRCNN RPN; README.md:8)
 * ---------------------------------------------------------------------------------------- */

void orc_anchor_base(const float* sizes, int n_sizes, const float* ratios, int n_ratios,
                     float* out_base) {
  int a = 0;
  for (int r = 0; r < n_ratios; ++r) {
    float hr = sqrtf(ratios[r]);
    for (int s = 0; s < n_sizes; ++s, ++a) {
      float w = sizes[s] / hr;
      float h = sizes[s] * hr;
      out_base[a * 4 + 0] = -w * 0.5f;
      out_base[a * 4 + 1] = -h * 0.5f;
      out_base[a * 4 + 2] = w * 0.5f;
      out_base[a * 4 + 3] = h * 0.5f;
    }
  }
}

/* level-major, then y, x, anchor. */
void orc_anchor_grid(int n_levels, const int* feat_h, const int* feat_w, const float* strides,
                     const float* base_anchors, int A, float offset, float* out) {
  int64_t o = 0;
  for (int l = 0; l < n_levels; ++l) {
    const float* base = base_anchors + (int64_t)l * A * 4;
    for (int y = 0; y < feat_h[l]; ++y)
      for (int x = 0; x < feat_w[l]; ++x) {
        float sx = ((float)x + offset) * strides[l];
        float sy = ((float)y + offset) * strides[l];
        for (int a = 0; a < A; ++a, ++o) {
          out[o * 4 + 0] = base[a * 4 + 0] + sx;
          out[o * 4 + 1] = base[a * 4 + 1] + sy;
          out[o * 4 + 2] = base[a * 4 + 2] + sx;
          out[o * 4 + 3] = base[a * 4 + 3] + sy;
        }
      }
  }
}

/* ------------------------------------------------------------------------------------------
 * pairwise IoU
 * ---------------------------------------------------------------------------------------- */

void orc_pairwise_iou(const float* a, const float* b, const int32_t* m_valid, int B, int N, int M,
                      float* out) {
  for (int bi = 0; bi < B; ++bi) {
    int mv = m_valid ? m_valid[bi] : M;
    for (int n = 0; n < N; ++n)
      for (int m = 0; m < M; ++m)
        out[((int64_t)bi * N + n) * M + m] =
            m < mv ? iou_xyxy(a + ((int64_t)bi * N + n) * 4, b + ((int64_t)bi * M + m) * 4)
                   : -1.0f;
  }
}

void orc_iou_assign(const float* a, const float* b, const int32_t* m_valid, int B, int N, int M,
                    float* max_iou, int32_t* argmax, float* gt_max_iou, int32_t* gt_argmax) {
  for (int bi = 0; bi < B; ++bi) {
    int mv = m_valid ? m_valid[bi] : M;
    if (gt_max_iou)
      for (int m = 0; m < M; ++m) {
        gt_max_iou[(int64_t)bi * M + m] = -1.0f;
        gt_argmax[(int64_t)bi * M + m] = -1;
      }
    for (int n = 0; n < N; ++n) {
      float best = -1.0f;
      int32_t bidx = -1;
      for (int m = 0; m < mv; ++m) {
        float v = iou_xyxy(a + ((int64_t)bi * N + n) * 4, b + ((int64_t)bi * M + m) * 4);
        if (v > best) { best = v; bidx = m; }
        if (gt_max_iou && v > gt_max_iou[(int64_t)bi * M + m]) {
          gt_max_iou[(int64_t)bi * M + m] = v;
          gt_argmax[(int64_t)bi * M + m] = n;
        }
      }
      max_iou[(int64_t)bi * N + n] = best;
      argmax[(int64_t)bi * N + n] = bidx;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * box delta codec
 * ---------------------------------------------------------------------------------------- */

#define ORC_DELTA_CLAMP 4.135166556742356f /* log(1000/16) */

void orc_box_encode(const float* src, const float* dst, int64_t n, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    const float* s = src + i * 4;
    const float* d = dst + i * 4;
    float sw = s[2] - s[0], sh = s[3] - s[1];
    float scx = s[0] + 0.5f * sw, scy = s[1] + 0.5f * sh;
    float dw = d[2] - d[0], dh = d[3] - d[1];
    float dcx = d[0] + 0.5f * dw, dcy = d[1] + 0.5f * dh;
    out[i * 4 + 0] = (dcx - scx) / sw;
    out[i * 4 + 1] = (dcy - scy) / sh;
    out[i * 4 + 2] = logf(dw / sw);
    out[i * 4 + 3] = logf(dh / sh);
  }
}

static void decode_one(const float* an, const float* dl, float img_h, float img_w, float* o) {
  float aw = an[2] - an[0], ah = an[3] - an[1];
  float acx = an[0] + 0.5f * aw, acy = an[1] + 0.5f * ah;
  float dw = fminf(dl[2], ORC_DELTA_CLAMP), dh = fminf(dl[3], ORC_DELTA_CLAMP);
  float cx = acx + dl[0] * aw, cy = acy + dl[1] * ah;
  float w = aw * expf(dw), h = ah * expf(dh);
  float x1 = cx - 0.5f * w, y1 = cy - 0.5f * h, x2 = cx + 0.5f * w, y2 = cy + 0.5f * h;
  if (img_w > 0.0f) {
    x1 = fminf(fmaxf(x1, 0.0f), img_w);
    x2 = fminf(fmaxf(x2, 0.0f), img_w);
    y1 = fminf(fmaxf(y1, 0.0f), img_h);
    y2 = fminf(fmaxf(y2, 0.0f), img_h);
  }
  o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2;
}

void orc_box_decode(const float* anchors, const float* deltas, int64_t n, float img_h,
                    float img_w, float* out) {
  for (int64_t i = 0; i < n; ++i) decode_one(anchors + i * 4, deltas + i * 4, img_h, img_w, out + i * 4);
}

/* ------------------------------------------------------------------------------------------
 * NMS
 * ---------------------------------------------------------------------------------------- */

typedef struct { uint32_t key; int32_t idx; } orc_sortrec;

/* score descending, ties -> lower index */
static int cmp_desc(const void* pa, const void* pb) {
  const orc_sortrec* a = (const orc_sortrec*)pa;
  const orc_sortrec* b = (const orc_sortrec*)pb;
  if (a->key != b->key) return a->key > b->key ? -1 : 1;
  return a->idx < b->idx ? -1 : (a->idx > b->idx ? 1 : 0);
}

/* Greedy NMS on one image; returns count. order[] receives candidate order (for callers). */
static int nms_one(const float* boxes, const float* scores, const int32_t* cls, int n_valid,
                   float score_thr, float iou_thr, int max_candidates, int max_out,
                   int32_t* keep) {
  orc_sortrec* rec = (orc_sortrec*)malloc(sizeof(orc_sortrec) * (size_t)(n_valid > 0 ? n_valid : 1));
  int nc = 0;
  for (int i = 0; i < n_valid; ++i)
    if (scores[i] > score_thr) { rec[nc].key = f2key(scores[i]); rec[nc].idx = i; ++nc; }
  qsort(rec, (size_t)nc, sizeof(orc_sortrec), cmp_desc);
  if (max_candidates > 0 && nc > max_candidates) nc = max_candidates;
  char* dead = (char*)calloc((size_t)(nc > 0 ? nc : 1), 1);
  int cnt = 0;
  for (int i = 0; i < nc && cnt < max_out; ++i) {
    if (dead[i]) continue;
    int bi = rec[i].idx;
    keep[cnt++] = bi;
    for (int j = i + 1; j < nc; ++j) {
      if (dead[j]) continue;
      int bj = rec[j].idx;
      if (cls && cls[bi] != cls[bj]) continue;
      if (iou_xyxy(boxes + (int64_t)bi * 4, boxes + (int64_t)bj * 4) > iou_thr) dead[j] = 1;
    }
  }
  free(dead);
  free(rec);
  return cnt;
}

void orc_nms_batched(const float* boxes, const float* scores, const int32_t* cls,
                     const int32_t* n_valid, int B, int N, float score_thr, float iou_thr,
                     int max_candidates, int max_out, int32_t* keep, int32_t* keep_count) {
  for (int b = 0; b < B; ++b) {
    int nv = n_valid ? n_valid[b] : N;
    if (nv > N) nv = N;
    int32_t* kb = keep + (int64_t)b * max_out;
    for (int i = 0; i < max_out; ++i) kb[i] = -1;
    keep_count[b] = nms_one(boxes + (int64_t)b * N * 4, scores + (int64_t)b * N,
                            cls ? cls + (int64_t)b * N : NULL, nv, score_thr, iou_thr,
                            max_candidates, max_out, kb);
  }
}

/* ------------------------------------------------------------------------------------------
 * RPN proposals (README.md:8)
 * ---------------------------------------------------------------------------------------- */

void orc_rpn_proposals(const float* logits, const float* deltas, const float* anchors, int B,
                       int NA, int n_levels, const int* level_sizes, int pre_nms_topk,
                       int post_nms_topk, float iou_thr, float min_size, float img_h, float img_w,
                       float* rois, float* roi_scores, int32_t* roi_index /* [B,post] or NULL */, int32_t* count) {
  int cap = 0;
  for (int l = 0; l < n_levels; ++l) cap += level_sizes[l] < pre_nms_topk ? level_sizes[l] : pre_nms_topk;
  float* cb = (float*)malloc(sizeof(float) * 4 * (size_t)(cap > 0 ? cap : 1));
  float* cs = (float*)malloc(sizeof(float) * (size_t)(cap > 0 ? cap : 1));
  int32_t* cl = (int32_t*)malloc(sizeof(int32_t) * (size_t)(cap > 0 ? cap : 1));
  int32_t* ci = (int32_t*)malloc(sizeof(int32_t) * (size_t)(cap > 0 ? cap : 1)); /* candidate -> anchor index */
  int32_t* keep = (int32_t*)malloc(sizeof(int32_t) * (size_t)(post_nms_topk > 0 ? post_nms_topk : 1));
  orc_sortrec* rec = (orc_sortrec*)malloc(sizeof(orc_sortrec) * (size_t)(NA > 0 ? NA : 1));
  for (int b = 0; b < B; ++b) {
    const float* lg = logits + (int64_t)b * NA;
    const float* dl = deltas + (int64_t)b * NA * 4;
    int nc = 0, off = 0;
    for (int l = 0; l < n_levels; ++l) {
      int n = level_sizes[l];
      for (int i = 0; i < n; ++i) { rec[i].key = f2key(lg[off + i]); rec[i].idx = off + i; }
      qsort(rec, (size_t)n, sizeof(orc_sortrec), cmp_desc);
      int k = n < pre_nms_topk ? n : pre_nms_topk;
      for (int i = 0; i < k; ++i) {
        int gi = rec[i].idx;
        float bx[4];
        decode_one(anchors + (int64_t)gi * 4, dl + (int64_t)gi * 4, img_h, img_w, bx);
        /* candidates stay in per-level top-k order; small boxes get score -inf (never pass) */
        int ok = (bx[2] - bx[0] >= min_size) && (bx[3] - bx[1] >= min_size);
        memcpy(cb + (int64_t)nc * 4, bx, sizeof(bx));
        cs[nc] = ok ? lg[gi] : -INFINITY;
        cl[nc] = l;
        ci[nc] = gi;
        ++nc;
      }
      off += n;
    }
    int cnt = nms_one(cb, cs, cl, nc, -INFINITY, iou_thr, 0, post_nms_topk, keep);
    count[b] = cnt;
    for (int i = 0; i < post_nms_topk; ++i) {
      float* r = rois + ((int64_t)b * post_nms_topk + i) * 4;
      if (i < cnt) {
        memcpy(r, cb + (int64_t)keep[i] * 4, 4 * sizeof(float));
        roi_scores[(int64_t)b * post_nms_topk + i] = sigmoidf_(cs[keep[i]]);
        if (roi_index) roi_index[(int64_t)b * post_nms_topk + i] = ci[keep[i]];
      } else {
        r[0] = r[1] = r[2] = r[3] = 0.0f;
        roi_scores[(int64_t)b * post_nms_topk + i] = 0.0f;
        if (roi_index) roi_index[(int64_t)b * post_nms_topk + i] = -1;
      }
    }
  }
  free(rec); free(keep); free(ci); free(cl); free(cs); free(cb);
}

/* ------------------------------------------------------------------------------------------
 * RoIAlign over pyramid levels (README.md:65,73)
 * ---------------------------------------------------------------------------------------- */

typedef struct orc_roi_levels {
  int n_levels;
  const float* feat[ORC_MAX_LEVELS];
  int h[ORC_MAX_LEVELS], w[ORC_MAX_LEVELS];
  float spatial_scale[ORC_MAX_LEVELS];
  int64_t sb[ORC_MAX_LEVELS], sc[ORC_MAX_LEVELS], sh[ORC_MAX_LEVELS], sw[ORC_MAX_LEVELS];
  int k_min;
  float canonical_scale;
  int canonical_level;
} orc_roi_levels;

static int roi_level(const float* roi, int k_min, int n_levels, float s0, int k0) {
  float area = (roi[3] - roi[1]) * (roi[4] - roi[2]);
  int lvl = 0;
  for (int k = k_min + 1; k < k_min + n_levels; ++k) {
    float t = s0 * ldexpf(1.0f, k - k0); /* exact power-of-two scaling */
    if (area >= t * t) ++lvl;
  }
  return lvl;
}

void orc_roi_level_map(const float* rois, int R, int k_min, int n_levels, float canonical_scale,
                       int canonical_level, int32_t* out_level) {
  for (int r = 0; r < R; ++r)
    out_level[r] = roi_level(rois + (int64_t)r * 5, k_min, n_levels, canonical_scale, canonical_level);
}

/* the four taps + weights of one bilinear sample; returns 0 when the sample is outside */
static int bilinear_taps(int H, int W, float y, float x, int* yl, int* xl, int* yh, int* xh,
                         float* w1, float* w2, float* w3, float* w4) {
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0;
  if (y <= 0.0f) y = 0.0f;
  if (x <= 0.0f) x = 0.0f;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
  if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
  float ly = y - (float)y_low, lx = x - (float)x_low;
  float hy = 1.0f - ly, hx = 1.0f - lx;
  *yl = y_low; *xl = x_low; *yh = y_high; *xh = x_high;
  *w1 = hy * hx; *w2 = hy * lx; *w3 = ly * hx; *w4 = ly * lx;
  return 1;
}

typedef struct { float start_w, start_h, bin_w, bin_h; int gh, gw; float count; } roi_geom;

static roi_geom roi_geometry(const float* roi, float scale, int P, int sampling_ratio, int aligned) {
  roi_geom g;
  float off = aligned ? 0.5f : 0.0f;
  g.start_w = roi[1] * scale - off;
  g.start_h = roi[2] * scale - off;
  float end_w = roi[3] * scale - off, end_h = roi[4] * scale - off;
  float rw = end_w - g.start_w, rh = end_h - g.start_h;
  if (!aligned) { rw = fmaxf(rw, 1.0f); rh = fmaxf(rh, 1.0f); }
  g.bin_h = rh / (float)P;
  g.bin_w = rw / (float)P;
  g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)P);
  g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)P);
  float c = (float)(g.gh * g.gw);
  g.count = c > 1.0f ? c : 1.0f;
  return g;
}

void orc_roi_align_fwd(const orc_roi_levels* lv, const float* rois, int R, int C, int P,
                       int sampling_ratio, int aligned, float* out, int64_t out_sr,
                       int64_t out_sc, int64_t out_sh, int64_t out_sw) {
  for (int r = 0; r < R; ++r) {
    const float* roi = rois + (int64_t)r * 5;
    int l = roi_level(roi, lv->k_min, lv->n_levels, lv->canonical_scale, lv->canonical_level);
    int bi = (int)roi[0];
    int H = lv->h[l], W = lv->w[l];
    roi_geom g = roi_geometry(roi, lv->spatial_scale[l], P, sampling_ratio, aligned);
    const float* f = lv->feat[l] + (int64_t)bi * lv->sb[l];
    for (int c = 0; c < C; ++c)
      for (int ph = 0; ph < P; ++ph)
        for (int pw = 0; pw < P; ++pw) {
          float acc = 0.0f;
          for (int iy = 0; iy < g.gh; ++iy) {
            float y = g.start_h + (float)ph * g.bin_h + ((float)iy + 0.5f) * g.bin_h / (float)g.gh;
            for (int ix = 0; ix < g.gw; ++ix) {
              float x = g.start_w + (float)pw * g.bin_w + ((float)ix + 0.5f) * g.bin_w / (float)g.gw;
              int yl, xl, yh, xh; float w1, w2, w3, w4;
              if (!bilinear_taps(H, W, y, x, &yl, &xl, &yh, &xh, &w1, &w2, &w3, &w4)) continue;
              const float* fc = f + (int64_t)c * lv->sc[l];
              float v1 = fc[yl * lv->sh[l] + xl * lv->sw[l]];
              float v2 = fc[yl * lv->sh[l] + xh * lv->sw[l]];
              float v3 = fc[yh * lv->sh[l] + xl * lv->sw[l]];
              float v4 = fc[yh * lv->sh[l] + xh * lv->sw[l]];
              acc += w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
            }
          }
          out[r * out_sr + c * out_sc + ph * out_sh + pw * out_sw] = acc / g.count;
        }
  }
}

/* Quantised RoI max-pooling (README.md:73 "-RoIPooling" variant), the public RoIPool definition:
 * corners rounded to cells (roundf: half away from zero), size = end - start + 1 (>= 1), bin rows
 * [floor(ph*bh), ceil((ph+1)*bh)) + start clipped to the map, first maximum in row-major order,
 * empty bin -> 0 with argmax -1.  argmax is y*W + x inside the RoI's level. */
typedef struct { int start_w, start_h; float bin_w, bin_h; } pool_geom;

static pool_geom pool_geometry(const float* roi, float scale, int P) {
  pool_geom g;
  g.start_w = (int)roundf(roi[1] * scale);
  g.start_h = (int)roundf(roi[2] * scale);
  int end_w = (int)roundf(roi[3] * scale), end_h = (int)roundf(roi[4] * scale);
  int rw = end_w - g.start_w + 1, rh = end_h - g.start_h + 1;
  if (rw < 1) rw = 1;
  if (rh < 1) rh = 1;
  g.bin_w = (float)rw / (float)P;
  g.bin_h = (float)rh / (float)P;
  return g;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void orc_roi_pool_fwd(const orc_roi_levels* lv, const float* rois, int R, int C, int P, float* out,
                      int32_t* argmax, int64_t out_sr, int64_t out_sc, int64_t out_sh, int64_t out_sw) {
  for (int r = 0; r < R; ++r) {
    const float* roi = rois + (int64_t)r * 5;
    int l = roi_level(roi, lv->k_min, lv->n_levels, lv->canonical_scale, lv->canonical_level);
    int bi = (int)roi[0];
    int H = lv->h[l], W = lv->w[l];
    pool_geom g = pool_geometry(roi, lv->spatial_scale[l], P);
    const float* f = lv->feat[l] + (int64_t)bi * lv->sb[l];
    for (int c = 0; c < C; ++c)
      for (int ph = 0; ph < P; ++ph) {
        int hs = clampi((int)floorf((float)ph * g.bin_h) + g.start_h, 0, H);
        int he = clampi((int)ceilf((float)(ph + 1) * g.bin_h) + g.start_h, 0, H);
        for (int pw = 0; pw < P; ++pw) {
          int ws = clampi((int)floorf((float)pw * g.bin_w) + g.start_w, 0, W);
          int we = clampi((int)ceilf((float)(pw + 1) * g.bin_w) + g.start_w, 0, W);
          int empty = (he <= hs) || (we <= ws);
          float best = empty ? 0.0f : -FLT_MAX;
          int arg = -1;
          const float* fc = f + (int64_t)c * lv->sc[l];
          for (int y = hs; y < he; ++y)
            for (int x = ws; x < we; ++x) {
              float v = fc[y * lv->sh[l] + x * lv->sw[l]];
              if (v > best) { best = v; arg = y * W + x; }
            }
          int64_t o = r * out_sr + c * out_sc + ph * out_sh + pw * out_sw;
          out[o] = best;
          argmax[o] = arg;
        }
      }
  }
}

void orc_roi_pool_bwd(const orc_roi_levels* grad_lv, const float* rois, int R, int C, int P,
                      const float* grad_out, const int32_t* argmax, int64_t out_sr, int64_t out_sc,
                      int64_t out_sh, int64_t out_sw) {
  for (int r = 0; r < R; ++r) {
    const float* roi = rois + (int64_t)r * 5;
    int l = roi_level(roi, grad_lv->k_min, grad_lv->n_levels, grad_lv->canonical_scale,
                      grad_lv->canonical_level);
    int bi = (int)roi[0];
    int W = grad_lv->w[l];
    float* f = (float*)grad_lv->feat[l] + (int64_t)bi * grad_lv->sb[l];
    for (int c = 0; c < C; ++c)
      for (int ph = 0; ph < P; ++ph)
        for (int pw = 0; pw < P; ++pw) {
          int64_t o = r * out_sr + c * out_sc + ph * out_sh + pw * out_sw;
          int a = argmax[o];
          if (a < 0) continue;
          f[(int64_t)c * grad_lv->sc[l] + (int64_t)(a / W) * grad_lv->sh[l] + (int64_t)(a % W) * grad_lv->sw[l]] +=
              grad_out[o];
        }
  }
}

/* grad_lv->feat[l] are WRITABLE gradient buffers (cast away const), zeroed by the caller */
void orc_roi_align_bwd(const orc_roi_levels* grad_lv, const float* rois, int R, int C, int P,
                       int sampling_ratio, int aligned, const float* grad_out, int64_t out_sr,
                       int64_t out_sc, int64_t out_sh, int64_t out_sw) {
  for (int r = 0; r < R; ++r) {
    const float* roi = rois + (int64_t)r * 5;
    int l = roi_level(roi, grad_lv->k_min, grad_lv->n_levels, grad_lv->canonical_scale,
                      grad_lv->canonical_level);
    int bi = (int)roi[0];
    int H = grad_lv->h[l], W = grad_lv->w[l];
    roi_geom g = roi_geometry(roi, grad_lv->spatial_scale[l], P, sampling_ratio, aligned);
    float* f = (float*)grad_lv->feat[l] + (int64_t)bi * grad_lv->sb[l];
    for (int c = 0; c < C; ++c)
      for (int ph = 0; ph < P; ++ph)
        for (int pw = 0; pw < P; ++pw) {
          float go = grad_out[r * out_sr + c * out_sc + ph * out_sh + pw * out_sw] / g.count;
          for (int iy = 0; iy < g.gh; ++iy) {
            float y = g.start_h + (float)ph * g.bin_h + ((float)iy + 0.5f) * g.bin_h / (float)g.gh;
            for (int ix = 0; ix < g.gw; ++ix) {
              float x = g.start_w + (float)pw * g.bin_w + ((float)ix + 0.5f) * g.bin_w / (float)g.gw;
              int yl, xl, yh, xh; float w1, w2, w3, w4;
              if (!bilinear_taps(H, W, y, x, &yl, &xl, &yh, &xh, &w1, &w2, &w3, &w4)) continue;
              float* fc = f + (int64_t)c * grad_lv->sc[l];
              fc[yl * grad_lv->sh[l] + xl * grad_lv->sw[l]] += go * w1;
              fc[yl * grad_lv->sh[l] + xh * grad_lv->sw[l]] += go * w2;
              fc[yh * grad_lv->sh[l] + xl * grad_lv->sw[l]] += go * w3;
              fc[yh * grad_lv->sh[l] + xh * grad_lv->sw[l]] += go * w4;
            }
          }
        }
  }
}

/* ------------------------------------------------------------------------------------------
 * YOLOv5 head ops (README.md:9)
 * ---------------------------------------------------------------------------------------- */

typedef struct orc_yolo_levels {
  int n_levels, na, no;
  float* pred[ORC_MAX_LEVELS];
  int h[ORC_MAX_LEVELS], w[ORC_MAX_LEVELS];
  float stride[ORC_MAX_LEVELS];
  float anchor_w[ORC_MAX_LEVELS][ORC_MAX_ANCHORS];
  float anchor_h[ORC_MAX_LEVELS][ORC_MAX_ANCHORS];
  int64_t sb[ORC_MAX_LEVELS], sa[ORC_MAX_LEVELS], sy[ORC_MAX_LEVELS], sx[ORC_MAX_LEVELS],
      so[ORC_MAX_LEVELS];
} orc_yolo_levels;

typedef struct orc_yolo_loss_cfg {
  float box_gain, obj_gain, cls_gain;
  float balance[ORC_MAX_LEVELS];
  float focal_gamma;  /* <= 0: plain BCE */
  float focal_alpha;  /* <= 0: no class-balance factor */
  int focal_terms;    /* bit 0: classification, bit 1: objectness */
} orc_yolo_loss_cfg;

/* Focal-modulated BCE-with-logits ("focalloss" rows, README.md:118-154; the public
 * FocalLoss-over-BCEWithLogits form) and its derivative w.r.t. the logit:
 *   L = af * q^gamma * bce, q = 1 - p_t, p_t = z p + (1-z)(1-p), af = z a + (1-z)(1-a)
 *   dL/dx = af * (q^gamma (p - z) + bce * dq^gamma/dx), dq^gamma/dx = -gamma q^(gamma-1) (2z-1) p (1-p) */
static float focal_bce(float x, float z, float gamma, float alpha, float* dldx) {
  float p = sigmoidf_(x);
  float bce = bce_logits(x, z);
  float q = 1.0f - (z * p + (1.0f - z) * (1.0f - p));
  float af = alpha > 0.0f ? z * alpha + (1.0f - z) * (1.0f - alpha) : 1.0f;
  float m = powf(q, gamma);
  float dm = q > 0.0f ? -gamma * (m / q) * ((2.0f * z - 1.0f) * (p * (1.0f - p))) : 0.0f;
  if (dldx) *dldx = af * (m * (p - z) + bce * dm);
  return af * (m * bce);
}

/* v6.1 decode; rows ordered level, anchor, y, x */
void orc_yolo_decode(const orc_yolo_levels* lv, int B, float* out) {
  int64_t n_total = 0;
  for (int l = 0; l < lv->n_levels; ++l) n_total += (int64_t)lv->na * lv->h[l] * lv->w[l];
  int64_t row0 = 0;
  for (int l = 0; l < lv->n_levels; ++l) {
    int H = lv->h[l], W = lv->w[l];
    for (int b = 0; b < B; ++b)
      for (int a = 0; a < lv->na; ++a)
        for (int y = 0; y < H; ++y)
          for (int x = 0; x < W; ++x) {
            const float* p = lv->pred[l] + b * lv->sb[l] + a * lv->sa[l] + y * lv->sy[l] + x * lv->sx[l];
            float* o = out + ((int64_t)b * n_total + row0 + ((int64_t)a * H + y) * W + x) * lv->no;
            float s0 = sigmoidf_(p[0]), s1 = sigmoidf_(p[lv->so[l]]);
            float s2 = sigmoidf_(p[2 * lv->so[l]]), s3 = sigmoidf_(p[3 * lv->so[l]]);
            o[0] = (s0 * 2.0f - 0.5f + (float)x) * lv->stride[l];
            o[1] = (s1 * 2.0f - 0.5f + (float)y) * lv->stride[l];
            o[2] = (s2 * 2.0f) * (s2 * 2.0f) * lv->anchor_w[l][a];
            o[3] = (s3 * 2.0f) * (s3 * 2.0f) * lv->anchor_h[l][a];
            for (int c = 4; c < lv->no; ++c) o[c] = sigmoidf_(p[c * lv->so[l]]);
          }
    row0 += (int64_t)lv->na * H * W;
  }
}

/* v6.1 build_targets form, fixed slots: slot = ((k*na + a)*B + b)*M + t */
void orc_yolo_assign(const orc_yolo_levels* lv, const float* gt_boxes, const int32_t* gt_cls,
                     const int32_t* m_valid, int B, int M, float anchor_t, int32_t* slot_cell,
                     float* slot_tbox, int32_t* slot_cls) {
  const int na = lv->na;
  const int64_t S = (int64_t)5 * na * B * M;
  for (int l = 0; l < lv->n_levels; ++l) {
    int H = lv->h[l], W = lv->w[l];
    float st = lv->stride[l];
    for (int k = 0; k < 5; ++k)
      for (int a = 0; a < na; ++a)
        for (int b = 0; b < B; ++b)
          for (int t = 0; t < M; ++t) {
            int64_t s = (((int64_t)k * na + a) * B + b) * M + t;
            int64_t o = (int64_t)l * S + s;
            slot_cell[o] = -1;
            slot_cls[o] = 0;
            slot_tbox[o * 4 + 0] = slot_tbox[o * 4 + 1] = slot_tbox[o * 4 + 2] = slot_tbox[o * 4 + 3] = 0.0f;
            if (t >= m_valid[b]) continue;
            const float* g = gt_boxes + ((int64_t)b * M + t) * 4;
            float gx = ((g[0] + g[2]) * 0.5f) / st, gy = ((g[1] + g[3]) * 0.5f) / st;
            float gw = (g[2] - g[0]) / st, gh = (g[3] - g[1]) / st;
            float aw = lv->anchor_w[l][a] / st, ah = lv->anchor_h[l][a] / st;
            float rw = gw / aw, rh = gh / ah;
            float m1 = fmaxf(rw, 1.0f / rw), m2 = fmaxf(rh, 1.0f / rh);
            if (!(fmaxf(m1, m2) < anchor_t)) continue;
            float fx = gx - floorf(gx), fy = gy - floorf(gy);
            float ix = (float)W - gx, iy = (float)H - gy;
            float fix = ix - floorf(ix), fiy = iy - floorf(iy);
            int sel; float ox, oy;
            switch (k) {
              case 0: sel = 1; ox = 0.0f; oy = 0.0f; break;
              case 1: sel = (fx < 0.5f) && (gx > 1.0f); ox = 0.5f; oy = 0.0f; break;
              case 2: sel = (fy < 0.5f) && (gy > 1.0f); ox = 0.0f; oy = 0.5f; break;
              case 3: sel = (fix < 0.5f) && (ix > 1.0f); ox = -0.5f; oy = 0.0f; break;
              default: sel = (fiy < 0.5f) && (iy > 1.0f); ox = 0.0f; oy = -0.5f; break;
            }
            if (!sel) continue;
            int gi = (int)(gx - ox), gj = (int)(gy - oy);
            gi = gi < 0 ? 0 : (gi > W - 1 ? W - 1 : gi);
            gj = gj < 0 ? 0 : (gj > H - 1 ? H - 1 : gj);
            slot_cell[o] = (int32_t)((((int64_t)b * na + a) * H + gj) * W + gi);
            slot_cls[o] = gt_cls[(int64_t)b * M + t];
            slot_tbox[o * 4 + 0] = gx - (float)gi;
            slot_tbox[o * 4 + 1] = gy - (float)gj;
            slot_tbox[o * 4 + 2] = gw;
            slot_tbox[o * 4 + 3] = gh;
          }
  }
}

/* CIoU of xyxy boxes with its gradient w.r.t. the first box (alpha held constant). */
#define ORC_EPS 1e-7f
#define ORC_4_PI2 0.4052847345693511f
static float ciou_xyxy(const float* p, const float* t, float* grad /* [4] or NULL */) {
  float w1 = p[2] - p[0], h1 = p[3] - p[1] + ORC_EPS;
  float w2 = t[2] - t[0], h2 = t[3] - t[1] + ORC_EPS;
  float iw = fminf(p[2], t[2]) - fmaxf(p[0], t[0]);
  float ih = fminf(p[3], t[3]) - fmaxf(p[1], t[1]);
  float iwc = iw > 0.0f ? iw : 0.0f, ihc = ih > 0.0f ? ih : 0.0f;
  float inter = iwc * ihc;
  float uni = w1 * h1 + w2 * h2 - inter + ORC_EPS;
  float iou = inter / uni;
  float cw = fmaxf(p[2], t[2]) - fminf(p[0], t[0]);
  float ch = fmaxf(p[3], t[3]) - fminf(p[1], t[1]);
  float c2 = cw * cw + ch * ch + ORC_EPS;
  float dx = t[0] + t[2] - p[0] - p[2], dy = t[1] + t[3] - p[1] - p[3];
  float rho2 = (dx * dx + dy * dy) * 0.25f;
  float at = atanf(w2 / h2) - atanf(w1 / h1);
  float v = ORC_4_PI2 * at * at;
  float alpha = v / (v - iou + (1.0f + ORC_EPS));
  float ciou = iou - (rho2 / c2 + v * alpha);
  if (grad) {
    /* d inter / d{x1,y1,x2,y2} */
    float diw_x1 = (iw > 0.0f && p[0] > t[0]) ? -1.0f : 0.0f;
    float diw_x2 = (iw > 0.0f && p[2] <= t[2]) ? 1.0f : 0.0f;
    float dih_y1 = (ih > 0.0f && p[1] > t[1]) ? -1.0f : 0.0f;
    float dih_y2 = (ih > 0.0f && p[3] <= t[3]) ? 1.0f : 0.0f;
    float dI[4] = {diw_x1 * ihc, dih_y1 * iwc, diw_x2 * ihc, dih_y2 * iwc};
    /* d (w1*h1) */
    float dA[4] = {-h1, -w1, h1, w1};
    /* enclosing box */
    float dcw_x1 = (p[0] <= t[0]) ? -1.0f : 0.0f, dcw_x2 = (p[2] >= t[2]) ? 1.0f : 0.0f;
    float dch_y1 = (p[1] <= t[1]) ? -1.0f : 0.0f, dch_y2 = (p[3] >= t[3]) ? 1.0f : 0.0f;
    float dC2[4] = {2.0f * cw * dcw_x1, 2.0f * ch * dch_y1, 2.0f * cw * dcw_x2, 2.0f * ch * dch_y2};
    float dR[4] = {-0.5f * dx, -0.5f * dy, -0.5f * dx, -0.5f * dy};
    /* v: d/dw1, d/dh1 */
    float den = w1 * w1 + h1 * h1;
    float dv_dw1 = -2.0f * ORC_4_PI2 * at * (h1 / den);
    float dv_dh1 = 2.0f * ORC_4_PI2 * at * (w1 / den);
    float dV[4] = {-dv_dw1, -dv_dh1, dv_dw1, dv_dh1};
    for (int i = 0; i < 4; ++i) {
      float dU = dA[i] - dI[i];
      float diou = (dI[i] * uni - inter * dU) / (uni * uni);
      float dpen = (dR[i] * c2 - rho2 * dC2[i]) / (c2 * c2);
      grad[i] = diou - dpen - alpha * dV[i];
    }
  }
  return ciou;
}

void orc_ciou_fwd(const float* pred, const float* target, int64_t P, float* ciou) {
  for (int64_t i = 0; i < P; ++i) ciou[i] = ciou_xyxy(pred + i * 4, target + i * 4, NULL);
}

void orc_ciou_bwd(const float* pred, const float* target, const float* grad_ciou, int64_t P,
                  float* grad_pred) {
  for (int64_t i = 0; i < P; ++i) {
    float g[4];
    ciou_xyxy(pred + i * 4, target + i * 4, g);
    for (int k = 0; k < 4; ++k) grad_pred[i * 4 + k] = grad_ciou[i] * g[k];
  }
}

static int64_t yolo_cells(const orc_yolo_levels* lv, int B, int l) {
  return (int64_t)B * lv->na * lv->h[l] * lv->w[l];
}

/* decode the box part of one positive: xywh in cell units -> xyxy, returns sigmoids for bwd */
static void yolo_pbox(const float* p, int64_t so, float aw_g, float ah_g, float* xyxy, float* sg) {
  for (int i = 0; i < 4; ++i) sg[i] = sigmoidf_(p[i * so]);
  float px = sg[0] * 2.0f - 0.5f, py = sg[1] * 2.0f - 0.5f;
  float pw = (sg[2] * 2.0f) * (sg[2] * 2.0f) * aw_g, ph = (sg[3] * 2.0f) * (sg[3] * 2.0f) * ah_g;
  xyxy[0] = px - pw * 0.5f; xyxy[1] = py - ph * 0.5f;
  xyxy[2] = px + pw * 0.5f; xyxy[3] = py + ph * 0.5f;
}

static const float* yolo_cell_ptr(const orc_yolo_levels* lv, int l, int32_t cell, int* a_out) {
  int H = lv->h[l], W = lv->w[l];
  int gi = cell % W; int r = cell / W;
  int gj = r % H; r /= H;
  int a = r % lv->na; int b = r / lv->na;
  *a_out = a;
  return lv->pred[l] + b * lv->sb[l] + a * lv->sa[l] + gj * lv->sy[l] + gi * lv->sx[l];
}

/* tobj: [sum_l cells_l] (level-major), acc: [n_levels*4], loss_out: [4] */
void orc_yolo_loss_fwd(const orc_yolo_levels* lv, int B, int M, const int32_t* slot_cell,
                       const float* slot_tbox, const int32_t* slot_cls,
                       const orc_yolo_loss_cfg* cfg, float* loss_out, float* tobj, float* acc) {
  const int nc = lv->no - 5;
  const int focal_cls = cfg->focal_gamma > 0.0f && (cfg->focal_terms & 1);
  const int focal_obj = cfg->focal_gamma > 0.0f && (cfg->focal_terms & 2);
  const int64_t S = (int64_t)5 * lv->na * B * M;
  int64_t toff = 0;
  double lbox = 0.0, lobj = 0.0, lcls = 0.0;
  for (int l = 0; l < lv->n_levels; ++l) {
    int64_t cells = yolo_cells(lv, B, l);
    float* to = tobj + toff;
    for (int64_t i = 0; i < cells; ++i) to[i] = 0.0f;
    double sb = 0.0, sc = 0.0, so_ = 0.0; int64_t n = 0;
    float st = lv->stride[l];
    for (int64_t s = 0; s < S; ++s) {
      int32_t cell = slot_cell[(int64_t)l * S + s];
      if (cell < 0) continue;
      int a; const float* p = yolo_cell_ptr(lv, l, cell, &a);
      float pb[4], sg[4];
      yolo_pbox(p, lv->so[l], lv->anchor_w[l][a] / st, lv->anchor_h[l][a] / st, pb, sg);
      const float* tb = slot_tbox + ((int64_t)l * S + s) * 4;
      float tx[4] = {tb[0] - tb[2] * 0.5f, tb[1] - tb[3] * 0.5f, tb[0] + tb[2] * 0.5f, tb[1] + tb[3] * 0.5f};
      float ci = ciou_xyxy(pb, tx, NULL);
      sb += (double)(1.0f - ci);
      ++n;
      float cl = ci > 0.0f ? ci : 0.0f;
      if (cl > to[cell]) to[cell] = cl;
      int32_t tc = slot_cls[(int64_t)l * S + s];
      if (nc > 1)
        for (int c = 0; c < nc; ++c) {
          float x = p[(5 + c) * lv->so[l]], z = c == tc ? 1.0f : 0.0f;
          sc += (double)(focal_cls ? focal_bce(x, z, cfg->focal_gamma, cfg->focal_alpha, NULL) : bce_logits(x, z));
        }
    }
    /* objectness over all cells */
    int H = lv->h[l], W = lv->w[l];
    for (int64_t cell = 0; cell < cells; ++cell) {
      int a; const float* p = yolo_cell_ptr(lv, l, (int32_t)cell, &a);
      so_ += (double)(focal_obj ? focal_bce(p[4 * lv->so[l]], to[cell], cfg->focal_gamma, cfg->focal_alpha, NULL)
                                : bce_logits(p[4 * lv->so[l]], to[cell]));
    }
    (void)H; (void)W;
    acc[l * 4 + 0] = (float)sb; acc[l * 4 + 1] = (float)n; acc[l * 4 + 2] = (float)sc; acc[l * 4 + 3] = (float)so_;
    if (n > 0) {
      lbox += sb / (double)n;
      if (nc > 1) lcls += sc / ((double)n * nc);
    }
    lobj += (so_ / (double)cells) * cfg->balance[l];
    toff += cells;
  }
  float fb = cfg->box_gain * (float)lbox, fo = cfg->obj_gain * (float)lobj, fc = cfg->cls_gain * (float)lcls;
  loss_out[0] = (float)B * (fb + fo + fc);
  loss_out[1] = fb; loss_out[2] = fo; loss_out[3] = fc;
}

/* dense gradient of `total` w.r.t. the raw head outputs, times grad_scale */
void orc_yolo_loss_bwd(const orc_yolo_levels* lv, const orc_yolo_levels* glv, int B, int M,
                       const int32_t* slot_cell, const float* slot_tbox, const int32_t* slot_cls,
                       const orc_yolo_loss_cfg* cfg, float grad_scale, const float* tobj,
                       const float* acc) {
  const int nc = lv->no - 5;
  const int focal_cls = cfg->focal_gamma > 0.0f && (cfg->focal_terms & 1);
  const int focal_obj = cfg->focal_gamma > 0.0f && (cfg->focal_terms & 2);
  const int64_t S = (int64_t)5 * lv->na * B * M;
  int64_t toff = 0;
  for (int l = 0; l < lv->n_levels; ++l) {
    int64_t cells = yolo_cells(lv, B, l);
    const float* to = tobj + toff;
    float st = lv->stride[l];
    float n = acc[l * 4 + 1];
    /* objectness + zero fill */
    float gobj = grad_scale * (float)B * cfg->obj_gain * cfg->balance[l] / (float)cells;
    for (int64_t cell = 0; cell < cells; ++cell) {
      int a; const float* p = yolo_cell_ptr(lv, l, (int32_t)cell, &a);
      float* g = (float*)yolo_cell_ptr(glv, l, (int32_t)cell, &a);
      for (int c = 0; c < lv->no; ++c) g[c * glv->so[l]] = 0.0f;
      float d;
      if (focal_obj) focal_bce(p[4 * lv->so[l]], to[cell], cfg->focal_gamma, cfg->focal_alpha, &d);
      else d = sigmoidf_(p[4 * lv->so[l]]) - to[cell];
      g[4 * glv->so[l]] = gobj * d;
    }
    if (n > 0.0f) {
      float gbox = grad_scale * (float)B * cfg->box_gain / n;
      float gcls = nc > 1 ? grad_scale * (float)B * cfg->cls_gain / (n * (float)nc) : 0.0f;
      for (int64_t s = 0; s < S; ++s) {
        int32_t cell = slot_cell[(int64_t)l * S + s];
        if (cell < 0) continue;
        int a; const float* p = yolo_cell_ptr(lv, l, cell, &a);
        float* g = (float*)yolo_cell_ptr(glv, l, cell, &a);
        float aw = lv->anchor_w[l][a] / st, ah = lv->anchor_h[l][a] / st;
        float pb[4], sg[4], gc[4];
        yolo_pbox(p, lv->so[l], aw, ah, pb, sg);
        const float* tb = slot_tbox + ((int64_t)l * S + s) * 4;
        float tx[4] = {tb[0] - tb[2] * 0.5f, tb[1] - tb[3] * 0.5f, tb[0] + tb[2] * 0.5f, tb[1] + tb[3] * 0.5f};
        ciou_xyxy(pb, tx, gc);
        /* loss = 1 - ciou -> dL/dxyxy = -gc; chain to (px,py,pw,ph) then to logits */
        float gpx = -(gc[0] + gc[2]), gpy = -(gc[1] + gc[3]);
        float gpw = -(gc[2] - gc[0]) * 0.5f, gph = -(gc[3] - gc[1]) * 0.5f;
        g[0 * glv->so[l]] += gbox * gpx * 2.0f * sg[0] * (1.0f - sg[0]);
        g[1 * glv->so[l]] += gbox * gpy * 2.0f * sg[1] * (1.0f - sg[1]);
        g[2 * glv->so[l]] += gbox * gpw * 8.0f * sg[2] * sg[2] * (1.0f - sg[2]) * aw;
        g[3 * glv->so[l]] += gbox * gph * 8.0f * sg[3] * sg[3] * (1.0f - sg[3]) * ah;
        int32_t tc = slot_cls[(int64_t)l * S + s];
        if (nc > 1)
          for (int c = 0; c < nc; ++c) {
            float x = p[(5 + c) * lv->so[l]], z = c == tc ? 1.0f : 0.0f, d;
            if (focal_cls) focal_bce(x, z, cfg->focal_gamma, cfg->focal_alpha, &d);
            else d = sigmoidf_(x) - z;
            g[(5 + c) * glv->so[l]] += gcls * d;
          }
      }
    }
    toff += cells;
  }
}

/* YOLOv5 post-processing on decoded rows */
void orc_yolo_nms(const float* dec, int B, int n, int no, float conf_thr, float iou_thr,
                  int max_nms, int max_det, float* det, int32_t* det_count) {
  float* bx = (float*)malloc(sizeof(float) * 4 * (size_t)n);
  float* sc = (float*)malloc(sizeof(float) * (size_t)n);
  int32_t* cl = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  int32_t* keep = (int32_t*)malloc(sizeof(int32_t) * (size_t)(max_det > 0 ? max_det : 1));
  for (int b = 0; b < B; ++b) {
    for (int i = 0; i < n; ++i) {
      const float* r = dec + ((int64_t)b * n + i) * no;
      float best = -1.0f; int bc = 0;
      for (int c = 5; c < no; ++c)
        if (r[c] > best) { best = r[c]; bc = c - 5; }
      float conf = r[4] * best;
      bx[i * 4 + 0] = r[0] - r[2] * 0.5f; bx[i * 4 + 1] = r[1] - r[3] * 0.5f;
      bx[i * 4 + 2] = r[0] + r[2] * 0.5f; bx[i * 4 + 3] = r[1] + r[3] * 0.5f;
      cl[i] = bc;
      sc[i] = (r[4] > conf_thr && conf > conf_thr) ? conf : -INFINITY;
    }
    int cnt = nms_one(bx, sc, cl, n, -INFINITY, iou_thr, max_nms, max_det, keep);
    det_count[b] = cnt;
    for (int i = 0; i < max_det; ++i) {
      float* d = det + ((int64_t)b * max_det + i) * 6;
      if (i < cnt) {
        int k = keep[i];
        d[0] = bx[k * 4]; d[1] = bx[k * 4 + 1]; d[2] = bx[k * 4 + 2]; d[3] = bx[k * 4 + 3];
        d[4] = sc[k]; d[5] = (float)cl[k];
      } else d[0] = d[1] = d[2] = d[3] = d[4] = d[5] = 0.0f;
    }
  }
  free(keep); free(cl); free(sc); free(bx);
}

/* ------------------------------------------------------------------------------------------
 * Weighted Boxes Fusion (README.md:19)
 * ---------------------------------------------------------------------------------------- */

typedef struct { float sx1, sy1, sx2, sy2, sconf; int n; int32_t label; float box[4]; int order; } orc_cluster;

static int cmp_cluster(const void* pa, const void* pb) {
  const float* a = (const float*)pa; const float* b = (const float*)pb; /* (score,label,order,...) */
  if (a[0] != b[0]) return a[0] > b[0] ? -1 : 1;
  if (a[1] != b[1]) return a[1] < b[1] ? -1 : 1;
  return a[2] < b[2] ? -1 : (a[2] > b[2] ? 1 : 0);
}

void orc_wbf(const float* boxes, const float* scores, const int32_t* labels,
             const float* model_weight, const int32_t* k_valid, int B, int K, float weight_sum,
             float iou_thr, float skip_thr, float* out, int32_t* out_count) {
  orc_sortrec* rec = (orc_sortrec*)malloc(sizeof(orc_sortrec) * (size_t)(K > 0 ? K : 1));
  orc_cluster* cl = (orc_cluster*)malloc(sizeof(orc_cluster) * (size_t)(K > 0 ? K : 1));
  float* srt = (float*)malloc(sizeof(float) * 8 * (size_t)(K > 0 ? K : 1));
  for (int b = 0; b < B; ++b) {
    const float* bx = boxes + (int64_t)b * K * 4;
    int kv = k_valid ? k_valid[b] : K;
    int nc = 0;
    for (int i = 0; i < kv; ++i) {
      if (scores[(int64_t)b * K + i] < skip_thr) continue;
      float conf = scores[(int64_t)b * K + i] * model_weight[(int64_t)b * K + i];
      rec[nc].key = f2key(conf); rec[nc].idx = i; ++nc;
    }
    qsort(rec, (size_t)nc, sizeof(orc_sortrec), cmp_desc);
    int ncl = 0;
    for (int i = 0; i < nc; ++i) {
      int bi = rec[i].idx;
      float conf = scores[(int64_t)b * K + bi] * model_weight[(int64_t)b * K + bi];
      int32_t lab = labels[(int64_t)b * K + bi];
      int best = -1; float best_iou = iou_thr;
      for (int c = 0; c < ncl; ++c) {
        if (cl[c].label != lab) continue;
        float v = iou_xyxy(cl[c].box, bx + (int64_t)bi * 4);
        if (v > best_iou) { best_iou = v; best = c; }
      }
      if (best < 0) {
        best = ncl++;
        cl[best].sx1 = cl[best].sy1 = cl[best].sx2 = cl[best].sy2 = cl[best].sconf = 0.0f;
        cl[best].n = 0; cl[best].label = lab; cl[best].order = best;
      }
      orc_cluster* c = &cl[best];
      c->sx1 += conf * bx[bi * 4 + 0]; c->sy1 += conf * bx[bi * 4 + 1];
      c->sx2 += conf * bx[bi * 4 + 2]; c->sy2 += conf * bx[bi * 4 + 3];
      c->sconf += conf; c->n += 1;
      c->box[0] = c->sx1 / c->sconf; c->box[1] = c->sy1 / c->sconf;
      c->box[2] = c->sx2 / c->sconf; c->box[3] = c->sy2 / c->sconf;
    }
    for (int c = 0; c < ncl; ++c) {
      float nn = (float)cl[c].n;
      float sc = (cl[c].sconf / nn) * fminf(weight_sum, nn) / weight_sum;
      srt[c * 8 + 0] = sc; srt[c * 8 + 1] = (float)cl[c].label; srt[c * 8 + 2] = (float)cl[c].order;
      srt[c * 8 + 3] = cl[c].box[0]; srt[c * 8 + 4] = cl[c].box[1];
      srt[c * 8 + 5] = cl[c].box[2]; srt[c * 8 + 6] = cl[c].box[3]; srt[c * 8 + 7] = 0.0f;
    }
    qsort(srt, (size_t)ncl, sizeof(float) * 8, cmp_cluster);
    out_count[b] = ncl;
    for (int i = 0; i < K; ++i) {
      float* o = out + ((int64_t)b * K + i) * 6;
      if (i < ncl) {
        o[0] = srt[i * 8 + 3]; o[1] = srt[i * 8 + 4]; o[2] = srt[i * 8 + 5]; o[3] = srt[i * 8 + 6];
        o[4] = srt[i * 8 + 0]; o[5] = srt[i * 8 + 1];
      } else o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = 0.0f;
    }
  }
  free(srt); free(cl); free(rec);
}

/* ------------------------------------------------------------------------------------------
 * Mosaic-4 on device (README.md:76-77,129-130 list Mosaic p=0.5/1.0 as the main augmentation; the
 * reference's augmentation code is not in the snapshot, so this is this repo's frozen convention,
 * DESIGN.md section 2 item 12).
 *   canvas 2S x 2S, centre (xc, yc) integer in [S/2, 3S/2]; tile k = (u >= xc) + 2 (v >= yc) shows source
 *   image src[k] translated by (ox, oy) = (xc - S | xc, yc - S | yc); uncovered canvas = fill;
 *   out(x, y) = ((c(2x,2y) + c(2x+1,2y)) + (c(2x,2y+1) + c(2x+1,2y+1))) * 0.25
 *   boxes: shift by (ox, oy), clip to the tile's visible rectangle, halve; keep iff w > min_wh, h > min_wh and
 *   clipped area / original area > min_area_ratio; output order tile-major then source order.
 *   apply[b] == 0: image src[b][0] and its boxes pass through unchanged.
 * images / out: float32, element strides (b, c, y, x) given.
 * ---------------------------------------------------------------------------------------- */
static void mosaic_tile(int k, int xc, int yc, int S, int* ox, int* oy, int* lx, int* ly, int* hx, int* hy) {
  *ox = (k & 1) ? xc : xc - S;
  *oy = (k & 2) ? yc : yc - S;
  *lx = (k & 1) ? xc : (*ox > 0 ? *ox : 0);
  *hx = (k & 1) ? ((*ox + S) < 2 * S ? (*ox + S) : 2 * S) : xc;
  *ly = (k & 2) ? yc : (*oy > 0 ? *oy : 0);
  *hy = (k & 2) ? ((*oy + S) < 2 * S ? (*oy + S) : 2 * S) : yc;
}

void orc_mosaic4(const float* images, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int B, int S,
                 const int32_t* src, const int32_t* center, const int32_t* apply, float fill, float* out,
                 int64_t ob, int64_t oc, int64_t oy_, int64_t ox_) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < 3; ++c)
      for (int y = 0; y < S; ++y)
        for (int x = 0; x < S; ++x) {
          float* o = out + b * ob + c * oc + y * oy_ + x * ox_;
          if (!apply[b]) {
            *o = images[(int64_t)src[b * 4] * sb + c * sc + y * sy + x * sx];
            continue;
          }
          int xc = center[b * 2], yc = center[b * 2 + 1];
          float p[4];
          for (int q = 0; q < 4; ++q) {
            int u = 2 * x + (q & 1), v = 2 * y + (q >> 1);
            int k = (u >= xc) + 2 * (v >= yc);
            int ox = (k & 1) ? xc : xc - S, oy = (k & 2) ? yc : yc - S;
            int su = u - ox, sv = v - oy;
            p[q] = (su >= 0 && su < S && sv >= 0 && sv < S)
                       ? images[(int64_t)src[b * 4 + k] * sb + c * sc + sv * sy + su * sx] : fill;
          }
          *o = ((p[0] + p[1]) + (p[2] + p[3])) * 0.25f;
        }
}

void orc_mosaic4_boxes(const float* gt_boxes, const int32_t* gt_cls, const int32_t* m_valid, int B, int M, int S,
                       const int32_t* src, const int32_t* center, const int32_t* apply, float min_wh,
                       float min_area_ratio, float* out_boxes, int32_t* out_cls, int32_t* out_count) {
  const int K = 4 * M;
  for (int b = 0; b < B; ++b) {
    int n = 0;
    float* ob = out_boxes + (int64_t)b * K * 4;
    int32_t* oc = out_cls + (int64_t)b * K;
    if (!apply[b]) {
      int s = src[b * 4];
      int mv = m_valid[s] < M ? m_valid[s] : M;
      for (int t = 0; t < mv; ++t) {
        for (int j = 0; j < 4; ++j) ob[n * 4 + j] = gt_boxes[((int64_t)s * M + t) * 4 + j];
        oc[n++] = gt_cls[(int64_t)s * M + t];
      }
    } else {
      int xc = center[b * 2], yc = center[b * 2 + 1];
      for (int k = 0; k < 4; ++k) {
        int ox, oy, lx, ly, hx, hy;
        mosaic_tile(k, xc, yc, S, &ox, &oy, &lx, &ly, &hx, &hy);
        int s = src[b * 4 + k];
        int mv = m_valid[s] < M ? m_valid[s] : M;
        for (int t = 0; t < mv; ++t) {
          const float* g = gt_boxes + ((int64_t)s * M + t) * 4;
          float x1 = g[0] + (float)ox, y1 = g[1] + (float)oy, x2 = g[2] + (float)ox, y2 = g[3] + (float)oy;
          float area0 = (x2 - x1) * (y2 - y1);
          x1 = fminf(fmaxf(x1, (float)lx), (float)hx); x2 = fminf(fmaxf(x2, (float)lx), (float)hx);
          y1 = fminf(fmaxf(y1, (float)ly), (float)hy); y2 = fminf(fmaxf(y2, (float)ly), (float)hy);
          x1 *= 0.5f; y1 *= 0.5f; x2 *= 0.5f; y2 *= 0.5f;
          float w = x2 - x1, h = y2 - y1;
          int keep = (w > min_wh) && (h > min_wh) && ((w * h) * 4.0f > min_area_ratio * area0);
          if (keep) {
            ob[n * 4 + 0] = x1; ob[n * 4 + 1] = y1; ob[n * 4 + 2] = x2; ob[n * 4 + 3] = y2;
            oc[n++] = gt_cls[(int64_t)s * M + t];
          }
        }
      }
    }
    out_count[b] = n;
    for (int i = n; i < K; ++i) {
      ob[i * 4 + 0] = ob[i * 4 + 1] = ob[i * 4 + 2] = ob[i * 4 + 3] = 0.0f;
      oc[i] = 0;
    }
  }
}

int orc_version(void) { return 1; }
