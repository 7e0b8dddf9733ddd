// roi_align.hip — RoIAlign (default) and quantised RoI max-pooling (the "-RoIPooling" variant)
// over pyramid levels, forward and backward
// (reference README.md:65,73 "RoIAlign" default, README.md:73 "FasterRCNN-FPNP2-RoIPooling";
// SURVEY.md §8a "RoIAlign over FPN levels").
//
// A gather: per RoI bin, sampling_ratio^2 bilinear samples x 4 taps x C channels.  The layout
// decision that matters on MI355X: features are consumed through explicit strides and the fast
// case is channels-last (sc == 1) — a wave then reads 64 consecutive channels of one tap as one
// 256-byte coalesced request, and neighbouring samples/bins of the same RoI hit the same lines
// in L1/L2 (a RoI touches ~ (roi_h*roi_w) * C * 4 bytes once from HBM; everything else is
// on-chip reuse).  Lanes = channels, so the tap coordinates and weights are wave-uniform scalar
// work and all vector work is loads + FMAs.
//
// One workgroup = one (RoI, pooled row ph); 256 lanes stride over channels.  The level of a
// RoI is found with exact area thresholds (no log2 rounding), identical to the oracle.
// Backward is the transposed scatter with float atomics (summation order not fixed).
#include "hd_common.h"

namespace {

__device__ __forceinline__ int roi_level_of(const float* roi, int k_min, int n_levels, float s0, int k0) {
  const float area = (roi[3] - roi[1]) * (roi[4] - roi[2]);
  int lvl = 0;
  for (int k = k_min + 1; k < k_min + n_levels; ++k) {
    const float t = ldexpf(s0, k - k0);  // s0 * 2^(k-k0), exact
    if (area >= t * t) ++lvl;
  }
  return lvl;
}

struct Taps {
  int yl, xl, yh, xh;
  float w1, w2, w3, w4;
  bool ok;
};

__device__ __forceinline__ Taps bilinear_taps(int H, int W, float y, float x) {
  Taps t;
  t.ok = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
  if (y <= 0.0f) y = 0.0f;
  if (x <= 0.0f) x = 0.0f;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) {
    y_high = y_low = H - 1;
    y = (float)y_low;
  } else {
    y_high = y_low + 1;
  }
  if (x_low >= W - 1) {
    x_high = x_low = W - 1;
    x = (float)x_low;
  } else {
    x_high = x_low + 1;
  }
  const float ly = y - (float)y_low, lx = x - (float)x_low;
  const float hy = 1.0f - ly, hx = 1.0f - lx;
  t.yl = y_low; t.xl = x_low; t.yh = y_high; t.xh = x_high;
  t.w1 = hy * hx; t.w2 = hy * lx; t.w3 = ly * hx; t.w4 = ly * lx;
  return t;
}

struct RoiGeom {
  float start_w, start_h, bin_w, bin_h, count;
  int gh, gw;
};

__device__ __forceinline__ RoiGeom roi_geometry(const float* roi, float scale, int P, int sampling_ratio,
                                                int aligned) {
  RoiGeom g;
  const float off = aligned ? 0.5f : 0.0f;
  g.start_w = roi[1] * scale - off;
  g.start_h = roi[2] * scale - off;
  const float end_w = roi[3] * scale - off, end_h = roi[4] * scale - off;
  float rw = end_w - g.start_w, rh = end_h - g.start_h;
  if (!aligned) {
    rw = fmaxf(rw, 1.0f);
    rh = fmaxf(rh, 1.0f);
  }
  g.bin_h = rh / (float)P;
  g.bin_w = rw / (float)P;
  g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)P);
  g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)P);
  const float c = (float)(g.gh * g.gw);
  g.count = c > 1.0f ? c : 1.0f;
  return g;
}

__global__ __launch_bounds__(256) void roi_level_map_kernel(const float* __restrict__ rois, int R, int k_min,
                                                            int n_levels, float s0, int k0,
                                                            int32_t* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < R) out[r] = roi_level_of(rois + (size_t)r * 5, k_min, n_levels, s0, k0);
}

// grid (R, P)
template <bool BWD>
__global__ __launch_bounds__(256) void roi_align_kernel(const hd_roi_levels lv, const float* __restrict__ rois, int C,
                                                        int P, int sampling_ratio, int aligned, float* out,
                                                        long long out_sr, long long out_sc, long long out_sh,
                                                        long long out_sw) {
  const int r = blockIdx.x, ph = blockIdx.y;
  const float* roi = rois + (size_t)r * 5;
  const int l = roi_level_of(roi, lv.k_min, lv.n_levels, lv.canonical_scale, lv.canonical_level);
  const int bi = (int)roi[0];
  const int H = lv.h[l], W = lv.w[l];
  const RoiGeom g = roi_geometry(roi, lv.spatial_scale[l], P, sampling_ratio, aligned);
  const long long sc = lv.sc[l], sh = lv.sh[l], sw = lv.sw[l];
  float* fbase = const_cast<float*>(lv.feat[l]) + (long long)bi * lv.sb[l];
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float* f = fbase + (long long)c * sc;
    for (int pw = 0; pw < P; ++pw) {
      float* o = out + r * out_sr + c * out_sc + ph * out_sh + pw * out_sw;
      float acc = 0.0f;
      float go = 0.0f;
      if (BWD) go = *o / g.count;
      for (int iy = 0; iy < g.gh; ++iy) {
        const float y = g.start_h + (float)ph * g.bin_h + ((float)iy + 0.5f) * g.bin_h / (float)g.gh;
        for (int ix = 0; ix < g.gw; ++ix) {
          const float x = g.start_w + (float)pw * g.bin_w + ((float)ix + 0.5f) * g.bin_w / (float)g.gw;
          const Taps t = bilinear_taps(H, W, y, x);
          if (!t.ok) continue;
          if (!BWD) {
            const float v1 = f[t.yl * sh + t.xl * sw], v2 = f[t.yl * sh + t.xh * sw];
            const float v3 = f[t.yh * sh + t.xl * sw], v4 = f[t.yh * sh + t.xh * sw];
            acc += t.w1 * v1 + t.w2 * v2 + t.w3 * v3 + t.w4 * v4;
          } else {
            atomicAdd(f + t.yl * sh + t.xl * sw, go * t.w1);
            atomicAdd(f + t.yl * sh + t.xh * sw, go * t.w2);
            atomicAdd(f + t.yh * sh + t.xl * sw, go * t.w3);
            atomicAdd(f + t.yh * sh + t.xh * sw, go * t.w4);
          }
        }
      }
      if (!BWD) *o = acc / g.count;
    }
  }
}

// ---- quantised RoI max-pooling -------------------------------------------------------------------
// Same mapping as RoIAlign: workgroup = (RoI, pooled row), lanes = channels, so the bin bounds are
// wave-uniform scalars and a wave reads 64 consecutive channels of one cell per request.
struct PoolGeom {
  int start_w, start_h;
  float bin_w, bin_h;
};

__device__ __forceinline__ PoolGeom pool_geometry(const float* roi, float scale, int P) {
  PoolGeom g;
  g.start_w = (int)roundf(roi[1] * scale);
  g.start_h = (int)roundf(roi[2] * scale);
  const int end_w = (int)roundf(roi[3] * scale), end_h = (int)roundf(roi[4] * scale);
  const int rw = max(end_w - g.start_w + 1, 1), rh = max(end_h - g.start_h + 1, 1);
  g.bin_w = (float)rw / (float)P;
  g.bin_h = (float)rh / (float)P;
  return g;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

__global__ __launch_bounds__(256) void roi_pool_fwd_kernel(const hd_roi_levels lv, const float* __restrict__ rois,
                                                           int C, int P, float* __restrict__ out,
                                                           int32_t* __restrict__ argmax, long long out_sr,
                                                           long long out_sc, long long out_sh, long long out_sw) {
  const int r = blockIdx.x, ph = blockIdx.y;
  const float* roi = rois + (size_t)r * 5;
  const int l = roi_level_of(roi, lv.k_min, lv.n_levels, lv.canonical_scale, lv.canonical_level);
  const int bi = (int)roi[0];
  const int H = lv.h[l], W = lv.w[l];
  const PoolGeom g = pool_geometry(roi, lv.spatial_scale[l], P);
  const long long sc = lv.sc[l], sh = lv.sh[l], sw = lv.sw[l];
  const float* fbase = lv.feat[l] + (long long)bi * lv.sb[l];
  const int hs = clampi((int)floorf((float)ph * g.bin_h) + g.start_h, 0, H);
  const int he = clampi((int)ceilf((float)(ph + 1) * g.bin_h) + g.start_h, 0, H);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float* f = fbase + (long long)c * sc;
    for (int pw = 0; pw < P; ++pw) {
      const int ws = clampi((int)floorf((float)pw * g.bin_w) + g.start_w, 0, W);
      const int we = clampi((int)ceilf((float)(pw + 1) * g.bin_w) + g.start_w, 0, W);
      const bool empty = (he <= hs) || (we <= ws);
      float best = empty ? 0.0f : -3.402823466e+38f;
      int arg = -1;
      for (int y = hs; y < he; ++y)
        for (int x = ws; x < we; ++x) {
          const float v = f[y * sh + x * sw];
          if (v > best) {
            best = v;
            arg = y * W + x;
          }
        }
      const long long o = r * out_sr + c * out_sc + ph * out_sh + pw * out_sw;
      out[o] = best;
      argmax[o] = arg;
    }
  }
}

// flat over [R, P, P, C-major-last]: every output element scatters one value
__global__ __launch_bounds__(256) void roi_pool_bwd_kernel(const hd_roi_levels lv, const float* __restrict__ rois,
                                                           int C, int P, const float* __restrict__ grad_out,
                                                           const int32_t* __restrict__ argmax, long long out_sr,
                                                           long long out_sc, long long out_sh, long long out_sw) {
  const int r = blockIdx.x, ph = blockIdx.y;
  const float* roi = rois + (size_t)r * 5;
  const int l = roi_level_of(roi, lv.k_min, lv.n_levels, lv.canonical_scale, lv.canonical_level);
  const int bi = (int)roi[0];
  const int W = lv.w[l];
  const long long sc = lv.sc[l], sh = lv.sh[l], sw = lv.sw[l];
  float* fbase = const_cast<float*>(lv.feat[l]) + (long long)bi * lv.sb[l];
  for (int c = threadIdx.x; c < C; c += blockDim.x)
    for (int pw = 0; pw < P; ++pw) {
      const long long o = r * out_sr + c * out_sc + ph * out_sh + pw * out_sw;
      const int a = argmax[o];
      if (a >= 0) atomicAdd(fbase + (long long)c * sc + (long long)(a / W) * sh + (long long)(a % W) * sw, grad_out[o]);
    }
}

inline hd_status check_roi_args(const hd_roi_levels* lv, const float* rois, int R, int C, int P, const float* out) {
  if (!lv) return HD_ERR_NULL;
  if (R < 0 || C <= 0 || P <= 0 || P > 65535 || lv->n_levels <= 0 || lv->n_levels > HD_MAX_LEVELS) return HD_ERR_SHAPE;
  if (R == 0) return HD_OK;
  if (!rois || !out) return HD_ERR_NULL;
  for (int l = 0; l < lv->n_levels; ++l) {
    if (!lv->feat[l]) return HD_ERR_NULL;
    if (lv->h[l] <= 0 || lv->w[l] <= 0) return HD_ERR_SHAPE;
  }
  return HD_OK;
}

}  // namespace

extern "C" hd_status hd_roi_level_map(const float* rois, int R, int k_min, int n_levels, float canonical_scale,
                                      int canonical_level, int32_t* out_level, hd_stream stream) {
  if (R < 0 || n_levels <= 0 || n_levels > HD_MAX_LEVELS) return HD_ERR_SHAPE;
  if (R == 0) return HD_OK;
  if (!rois || !out_level) return HD_ERR_NULL;
  hipLaunchKernelGGL(roi_level_map_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, rois, R, k_min,
                     n_levels, canonical_scale, canonical_level, out_level);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_roi_align_fwd(const hd_roi_levels* levels, const float* rois, int R, int C, int P,
                                      int sampling_ratio, int aligned, float* out, int64_t out_sr, int64_t out_sc,
                                      int64_t out_sh, int64_t out_sw, hd_stream stream) {
  hd_status st = check_roi_args(levels, rois, R, C, P, out);
  if (st != HD_OK || R == 0) return st;
  const int threads = C >= 256 ? 256 : (C > 128 ? 256 : (C > 64 ? 128 : 64));
  hipLaunchKernelGGL((roi_align_kernel<false>), dim3(R, P), dim3(threads), 0, (hipStream_t)stream, *levels, rois, C,
                     P, sampling_ratio, aligned, out, (long long)out_sr, (long long)out_sc, (long long)out_sh,
                     (long long)out_sw);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_roi_align_bwd(const hd_roi_levels* grad_levels, const float* rois, int R, int C, int P,
                                      int sampling_ratio, int aligned, const float* grad_out, int64_t out_sr,
                                      int64_t out_sc, int64_t out_sh, int64_t out_sw, hd_stream stream) {
  hd_status st = check_roi_args(grad_levels, rois, R, C, P, grad_out);
  if (st != HD_OK || R == 0) return st;
  const int threads = C >= 256 ? 256 : (C > 128 ? 256 : (C > 64 ? 128 : 64));
  hipLaunchKernelGGL((roi_align_kernel<true>), dim3(R, P), dim3(threads), 0, (hipStream_t)stream, *grad_levels, rois,
                     C, P, sampling_ratio, aligned, const_cast<float*>(grad_out), (long long)out_sr,
                     (long long)out_sc, (long long)out_sh, (long long)out_sw);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_roi_pool_fwd(const hd_roi_levels* levels, const float* rois, int R, int C, int P, float* out,
                                     int32_t* argmax, int64_t out_sr, int64_t out_sc, int64_t out_sh,
                                     int64_t out_sw, hd_stream stream) {
  hd_status st = check_roi_args(levels, rois, R, C, P, out);
  if (st != HD_OK || R == 0) return st;
  if (!argmax) return HD_ERR_NULL;
  const int threads = C > 128 ? 256 : (C > 64 ? 128 : 64);
  hipLaunchKernelGGL(roi_pool_fwd_kernel, dim3(R, P), dim3(threads), 0, (hipStream_t)stream, *levels, rois, C, P, out,
                     argmax, (long long)out_sr, (long long)out_sc, (long long)out_sh, (long long)out_sw);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_roi_pool_bwd(const hd_roi_levels* grad_levels, const float* rois, int R, int C, int P,
                                     const float* grad_out, const int32_t* argmax, int64_t out_sr, int64_t out_sc,
                                     int64_t out_sh, int64_t out_sw, hd_stream stream) {
  hd_status st = check_roi_args(grad_levels, rois, R, C, P, grad_out);
  if (st != HD_OK || R == 0) return st;
  if (!argmax) return HD_ERR_NULL;
  const int threads = C > 128 ? 256 : (C > 64 ? 128 : 64);
  hipLaunchKernelGGL(roi_pool_bwd_kernel, dim3(R, P), dim3(threads), 0, (hipStream_t)stream, *grad_levels, rois, C, P,
                     grad_out, argmax, (long long)out_sr, (long long)out_sc, (long long)out_sh, (long long)out_sw);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}
