// mosaic.hip — Mosaic-4 augmentation on device (reference README.md:76-77,129-130: "Mosaic p=0.5 / 1.0" is the
// main augmentation of every result table; SURVEY.md §8f rank 3 "the step before the hot path").  At ~2000
// images/s per GPU a host-side mosaic would bound the step, so the batch is recombined in HBM.
// Convention: DESIGN.md §2 item 12 (the reference's own augmentation code is not in the snapshot).
//
//   mosaic4_image_kernel   one lane per output element, enumerated in the output's memory order (channels-last
//                          or planar), so stores are coalesced; each element averages 4 canvas pixels, each of
//                          which is one source pixel or the fill value.  Every source pixel is needed by exactly
//                          one output element, so the kernel is a pure HBM stream: 4 reads + 1 write per element.
//   mosaic4_boxes_kernel   one wave per output image: shift / clip / halve / filter the boxes of the 4 source
//                          images and compact them in (tile, source index) order with ballot prefix sums.
#include "hd_common.h"

namespace {

struct MosaicImg {
  const float* in;
  float* out;
  long long sb, sc, sy, sx, ob, oc, oy, ox;
};

__global__ __launch_bounds__(256) void mosaic4_image_kernel(const MosaicImg t, int B, int S,
                                                            const int32_t* __restrict__ src,
                                                            const int32_t* __restrict__ center,
                                                            const int32_t* __restrict__ apply, float fill,
                                                            int c_fastest) {
  const int b = blockIdx.y;
  const unsigned per_img = 3u * (unsigned)S * (unsigned)S;
  const int xc = center[b * 2], yc = center[b * 2 + 1];
  const int ap = apply[b];
  const int s0 = src[b * 4], s1 = src[b * 4 + 1], s2 = src[b * 4 + 2], s3 = src[b * 4 + 3];
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < per_img; e += gridDim.x * blockDim.x) {
    int c, y, x;
    if (c_fastest) {  // memory order (y, x, c)
      c = (int)(e % 3u);
      const unsigned r = e / 3u;
      x = (int)(r % (unsigned)S);
      y = (int)(r / (unsigned)S);
    } else {          // memory order (c, y, x)
      x = (int)(e % (unsigned)S);
      const unsigned r = e / (unsigned)S;
      y = (int)(r % (unsigned)S);
      c = (int)(r / (unsigned)S);
    }
    float v;
    if (!ap) {
      v = t.in[(long long)s0 * t.sb + c * t.sc + y * t.sy + x * t.sx];
    } else {
      float p[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int u = 2 * x + (q & 1), w = 2 * y + (q >> 1);
        const bool right = u >= xc, low = w >= yc;
        const int su = u - (right ? xc : xc - S), sv = w - (low ? yc : yc - S);
        const int s = low ? (right ? s3 : s2) : (right ? s1 : s0);
        p[q] = (su >= 0 && su < S && sv >= 0 && sv < S) ? t.in[(long long)s * t.sb + c * t.sc + sv * t.sy + su * t.sx]
                                                        : fill;
      }
      v = ((p[0] + p[1]) + (p[2] + p[3])) * 0.25f;
    }
    t.out[(long long)b * t.ob + c * t.oc + y * t.oy + x * t.ox] = v;
  }
}

// one 64-lane workgroup per output image
__global__ __launch_bounds__(64) void mosaic4_boxes_kernel(const float4* __restrict__ gt_boxes,
                                                           const int32_t* __restrict__ gt_cls,
                                                           const int32_t* __restrict__ m_valid, int M, int S,
                                                           const int32_t* __restrict__ src,
                                                           const int32_t* __restrict__ center,
                                                           const int32_t* __restrict__ apply, float min_wh,
                                                           float min_area_ratio, float4* __restrict__ out_boxes,
                                                           int32_t* __restrict__ out_cls,
                                                           int32_t* __restrict__ out_count) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int K = 4 * M;
  float4* ob = out_boxes + (size_t)b * K;
  int32_t* oc = out_cls + (size_t)b * K;
  const int xc = center[b * 2], yc = center[b * 2 + 1];
  const int ap = apply[b];
  int n = 0;  // wave-uniform running count
  const int tiles = ap ? 4 : 1;
  for (int k = 0; k < tiles; ++k) {
    const int s = src[b * 4 + k];
    const int mv = min(m_valid[s], M);
    const int ox = (k & 1) ? xc : xc - S, oy = (k & 2) ? yc : yc - S;
    const float lx = (float)((k & 1) ? xc : max(ox, 0)), hx = (float)((k & 1) ? min(ox + S, 2 * S) : xc);
    const float ly = (float)((k & 2) ? yc : max(oy, 0)), hy = (float)((k & 2) ? min(oy + S, 2 * S) : yc);
    for (int t0 = 0; t0 < mv; t0 += 64) {
      const int t = t0 + lane;
      bool keep = false;
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      int cl = 0;
      if (t < mv) {
        g = gt_boxes[(size_t)s * M + t];
        cl = gt_cls[(size_t)s * M + t];
        if (!ap) {
          keep = true;
        } else {
          float x1 = g.x + (float)ox, y1 = g.y + (float)oy, x2 = g.z + (float)ox, y2 = g.w + (float)oy;
          const float area0 = (x2 - x1) * (y2 - y1);
          x1 = fminf(fmaxf(x1, lx), hx); x2 = fminf(fmaxf(x2, lx), hx);
          y1 = fminf(fmaxf(y1, ly), hy); y2 = fminf(fmaxf(y2, ly), hy);
          x1 *= 0.5f; y1 *= 0.5f; x2 *= 0.5f; y2 *= 0.5f;
          const float w = x2 - x1, h = y2 - y1;
          keep = (w > min_wh) && (h > min_wh) && ((w * h) * 4.0f > min_area_ratio * area0);
          g = make_float4(x1, y1, x2, y2);
        }
      }
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
        ob[pos] = g;
        oc[pos] = cl;
      }
      n += __popcll(m);
    }
  }
  if (lane == 0) out_count[b] = n;
  for (int i = n + lane; i < K; i += 64) {
    ob[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    oc[i] = 0;
  }
}

}  // namespace

extern "C" hd_status hd_mosaic4(const float* images, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int B, int S,
                                const int32_t* src, const int32_t* center, const int32_t* apply, float fill,
                                float* out, int64_t ob, int64_t oc, int64_t oy, int64_t ox, hd_stream stream) {
  if (B < 0 || S <= 0 || (S & 1) || B > 65535 || (long long)S * S * 3 >= (1ll << 31)) return HD_ERR_SHAPE;
  if (B == 0) return HD_OK;
  if (!images || !src || !center || !apply || !out) return HD_ERR_NULL;
  if (images == out) return HD_ERR_SHAPE;  // every output element reads other images of the batch
  MosaicImg t{images, out, sb, sc, sy, sx, ob, oc, oy, ox};
  const unsigned per_img = 3u * (unsigned)S * (unsigned)S;
  unsigned gx = (per_img + 256u * 4u - 1u) / (256u * 4u);  // ~4 elements per lane
  if (gx > 4096u) gx = 4096u;
  hipLaunchKernelGGL(mosaic4_image_kernel, dim3(gx, (unsigned)B), dim3(256), 0, (hipStream_t)stream, t, B, S, src,
                     center, apply, fill, oc == 1 ? 1 : 0);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_mosaic4_boxes(const float* gt_boxes, const int32_t* gt_cls, const int32_t* m_valid, int B,
                                      int M, int S, const int32_t* src, const int32_t* center, const int32_t* apply,
                                      float min_wh, float min_area_ratio, float* out_boxes, int32_t* out_cls,
                                      int32_t* out_count, hd_stream stream) {
  if (B < 0 || M < 0 || S <= 0) return HD_ERR_SHAPE;
  if (B == 0) return HD_OK;
  if (!m_valid || !src || !center || !apply || !out_count) return HD_ERR_NULL;
  if (M > 0 && (!gt_boxes || !gt_cls || !out_boxes || !out_cls)) return HD_ERR_NULL;
  if (M > 0 && (!hd_aligned16(gt_boxes) || !hd_aligned16(out_boxes))) return HD_ERR_ALIGN;
  hipLaunchKernelGGL(mosaic4_boxes_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(gt_boxes), gt_cls, m_valid, M, S, src, center, apply, min_wh,
                     min_area_ratio, reinterpret_cast<float4*>(out_boxes), out_cls, out_count);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}
