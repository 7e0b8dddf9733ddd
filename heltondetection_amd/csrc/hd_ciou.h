// hd_ciou.h — device CIoU (with gradient) and delta decode shared by codec.hip, yolo.hip, rpn.hip.
// Term-by-term the arithmetic of oracle/hd_oracle.c: ciou_xyxy() / decode_one().
#pragma once
#include "hd_common.h"

#define HD_CIOU_EPS 1e-7f
#define HD_4_PI2 0.4052847345693511f

// Returns CIoU(p, t); when GRAD, g[0..3] = d ciou / d (p.x1, p.y1, p.x2, p.y2), alpha constant.
template <bool GRAD>
__device__ __forceinline__ float hd_ciou(const float4 p, const float4 t, float* g) {
  const float w1 = p.z - p.x, h1 = p.w - p.y + HD_CIOU_EPS;
  const float w2 = t.z - t.x, h2 = t.w - t.y + HD_CIOU_EPS;
  const float iw = fminf(p.z, t.z) - fmaxf(p.x, t.x);
  const float ih = fminf(p.w, t.w) - fmaxf(p.y, t.y);
  const float iwc = iw > 0.0f ? iw : 0.0f, ihc = ih > 0.0f ? ih : 0.0f;
  const float inter = iwc * ihc;
  const float uni = w1 * h1 + w2 * h2 - inter + HD_CIOU_EPS;
  const float iou = inter / uni;
  const float cw = fmaxf(p.z, t.z) - fminf(p.x, t.x);
  const float ch = fmaxf(p.w, t.w) - fminf(p.y, t.y);
  const float c2 = cw * cw + ch * ch + HD_CIOU_EPS;
  const float dx = t.x + t.z - p.x - p.z, dy = t.y + t.w - p.y - p.w;
  const float rho2 = (dx * dx + dy * dy) * 0.25f;
  const float at = atanf(w2 / h2) - atanf(w1 / h1);
  const float v = HD_4_PI2 * at * at;
  const float alpha = v / (v - iou + (1.0f + HD_CIOU_EPS));
  const float ciou = iou - (rho2 / c2 + v * alpha);
  if (GRAD) {
    const float dI[4] = {((iw > 0.0f && p.x > t.x) ? -1.0f : 0.0f) * ihc,
                         ((ih > 0.0f && p.y > t.y) ? -1.0f : 0.0f) * iwc,
                         ((iw > 0.0f && p.z <= t.z) ? 1.0f : 0.0f) * ihc,
                         ((ih > 0.0f && p.w <= t.w) ? 1.0f : 0.0f) * iwc};
    const float dA[4] = {-h1, -w1, h1, w1};
    const float dC2[4] = {2.0f * cw * ((p.x <= t.x) ? -1.0f : 0.0f), 2.0f * ch * ((p.y <= t.y) ? -1.0f : 0.0f),
                          2.0f * cw * ((p.z >= t.z) ? 1.0f : 0.0f), 2.0f * ch * ((p.w >= t.w) ? 1.0f : 0.0f)};
    const float dR[4] = {-0.5f * dx, -0.5f * dy, -0.5f * dx, -0.5f * dy};
    const float den = w1 * w1 + h1 * h1;
    const float dv_dw1 = -2.0f * HD_4_PI2 * at * (h1 / den);
    const float dv_dh1 = 2.0f * HD_4_PI2 * at * (w1 / den);
    const float dV[4] = {-dv_dw1, -dv_dh1, dv_dw1, dv_dh1};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float dU = dA[i] - dI[i];
      const float diou = (dI[i] * uni - inter * dU) / (uni * uni);
      const float dpen = (dR[i] * c2 - rho2 * dC2[i]) / (c2 * c2);
      g[i] = diou - dpen - alpha * dV[i];
    }
  }
  return ciou;
}

__device__ __forceinline__ float4 hd_delta_decode(const float4 an, const float4 dl, float img_h, float img_w) {
  const float aw = an.z - an.x, ah = an.w - an.y;
  const float acx = an.x + 0.5f * aw, acy = an.y + 0.5f * ah;
  const float dw = fminf(dl.z, 4.135166556742356f), dh = fminf(dl.w, 4.135166556742356f);
  const float cx = acx + dl.x * aw, cy = acy + dl.y * ah;
  const float w = aw * expf(dw), h = ah * expf(dh);
  float x1 = cx - 0.5f * w, y1 = cy - 0.5f * h, x2 = cx + 0.5f * w, y2 = cy + 0.5f * h;
  if (img_w > 0.0f) {
    x1 = fminf(fmaxf(x1, 0.0f), img_w);
    x2 = fminf(fmaxf(x2, 0.0f), img_w);
    y1 = fminf(fmaxf(y1, 0.0f), img_h);
    y2 = fminf(fmaxf(y2, 0.0f), img_h);
  }
  return make_float4(x1, y1, x2, y2);
}
