// roi_align.hip — RoIAlign (default) and quantised RoI max-pooling (the "-RoIPooling" variant)
// over pyramid levels, forward and backward
// (reference README.md:65,73 "RoIAlign" default, README.md:73 "FasterRCNN-FPNP2-RoIPooling";
// SURVEY.md §8a "RoIAlign over FPN levels": f32, or bf16 features with f32 accumulation).
//
// A gather: per RoI bin, sampling_ratio^2 bilinear samples x 4 taps x C channels.  Bilinear weights
// factor into a row part and a column part, and so does the sample grid of a bin, so
//     out[ph][pw][c] = 1/count * sum_{(y, wy) in rows(ph)} sum_{(x, wx) in cols(pw)} wy * wx * f[y][x][c]
// (a sample outside [-1, size] zeroes its own row or column weight — exactly the "contributes 0" rule).
//
// Fast path (channels-last features, sampling_ratio 1 or 2), roi_align_fwd_cl_kernel:
//   * one workgroup per RoI; a thread owns (pooled row ph, 16 bytes of channels: 8 bf16 or 4 f32), so a
//     wave instruction moves 1 KiB instead of 256 B and the rows of a RoI run side by side;
//   * the <= 2*SR row taps of each ph and the 2 column taps of every column sample are computed once
//     per RoI by the first lanes and parked in LDS (the loop reads them back as broadcasts);
//   * a thread walks its row's column samples left to right keeping the two most recent COLUMN VALUES
//     colval(x) = sum_y wy f[y][x] in registers: consecutive samples are < 1 px apart for every RoI the
//     level map sends to a level (7..14 px wide there), so each feature pixel under the RoI row is
//     loaded once per ph instead of once per tap (28 column taps -> roi_w + 2 column loads);
//   * blockIdx -> RoI is XCD-aware: blocks b and b + 8 share an XCD (round-robin dispatch), so XCD k
//     is given the contiguous RoI range [k*R/8, (k+1)*R/8) = the RoIs of two images, whose feature maps
//     then stay in that XCD's L2 (speed only; nothing depends on the placement).
// Backward fast path, roi_align_bwd_cl_kernel: workgroup = (RoI, ph), lane = channel (so every atomic
// wave instruction covers 256 contiguous bytes, the shape the memory-side atomic units run at full
// rate for), same separable walk transposed: the column gradients are accumulated in registers while the
// samples move right and each finished column is flushed with one float atomic per distinct row tap:
// rows x (roi_w + 2) atomics per (RoI, ph) instead of 16 per bin.
// Everything else (NCHW features, adaptive sampling_ratio <= 0, sampling_ratio > 2, odd strides) runs
// roi_align_kernel: one workgroup per (RoI, ph), lanes = channels, explicit strides.
// The level of a RoI is found with exact area thresholds (no log2 rounding), identical to the oracle.
// Backward uses float atomics (summation order not fixed).
#include "hd_common.h"
#include "hd_vec.h"

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

// ---- element access of the generic (any-stride) kernels ------------------------------------------
__device__ __forceinline__ float ld_elem(const float* p) { return *p; }
__device__ __forceinline__ float ld_elem(const __hip_bfloat16* p) {
  return __uint_as_float((uint32_t)(*reinterpret_cast<const uint16_t*>(p)) << 16);
}
__device__ __forceinline__ void st_elem(float* p, float v) { *p = v; }
__device__ __forceinline__ void st_elem(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }

// Generic forward: grid (R, P), lanes = channels, explicit strides, any sampling ratio.
template <typename TF, typename TO>
__global__ __launch_bounds__(256) void roi_align_fwd_kernel(const hd_roi_levels lv, const float* __restrict__ rois, int C,
                                                            int P, int sampling_ratio, int aligned, TO* __restrict__ out,
                                                            long long out_sr, long long out_sc, long long out_sh,
                                                            long long out_sw) {
  const int r = blockIdx.x, ph = blockIdx.y;
  const float* roi = rois + (size_t)r * 5;
  const int l = roi_level_of(roi, lv.k_min, lv.n_levels, lv.canonical_scale, lv.canonical_level);
  const int bi = (int)roi[0];
  const int H = lv.h[l], W = lv.w[l];
  const RoiGeom g = roi_geometry(roi, lv.spatial_scale[l], P, sampling_ratio, aligned);
  const long long sc = lv.sc[l], sh = lv.sh[l], sw = lv.sw[l];
  const TF* fbase = reinterpret_cast<const TF*>(lv.feat[l]) + (long long)bi * lv.sb[l];
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const TF* f = fbase + (long long)c * sc;
    for (int pw = 0; pw < P; ++pw) {
      float acc = 0.0f;
      for (int iy = 0; iy < g.gh; ++iy) {
        const float y = g.start_h + (float)ph * g.bin_h + ((float)iy + 0.5f) * g.bin_h / (float)g.gh;
        for (int ix = 0; ix < g.gw; ++ix) {
          const float x = g.start_w + (float)pw * g.bin_w + ((float)ix + 0.5f) * g.bin_w / (float)g.gw;
          const Taps t = bilinear_taps(H, W, y, x);
          if (!t.ok) continue;
          const float v1 = ld_elem(f + t.yl * sh + t.xl * sw), v2 = ld_elem(f + t.yl * sh + t.xh * sw);
          const float v3 = ld_elem(f + t.yh * sh + t.xl * sw), v4 = ld_elem(f + t.yh * sh + t.xh * sw);
          acc += t.w1 * v1 + t.w2 * v2 + t.w3 * v3 + t.w4 * v4;
        }
      }
      st_elem(out + r * out_sr + c * out_sc + ph * out_sh + pw * out_sw, acc / g.count);
    }
  }
}

// Generic backward: float atomics into f32 gradient maps, grad_out f32 or bf16.
template <typename TG>
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(const hd_roi_levels lv, const float* __restrict__ rois, int C,
                                                            int P, int sampling_ratio, int aligned,
                                                            const TG* __restrict__ grad_out, long long out_sr,
                                                            long long out_sc, long long out_sh, long long out_sw) {
  const int r = blockIdx.x, ph = blockIdx.y;
  const float* roi = rois + (size_t)r * 5;
  const int l = roi_level_of(roi, lv.k_min, lv.n_levels, lv.canonical_scale, lv.canonical_level);
  const int bi = (int)roi[0];
  const int H = lv.h[l], W = lv.w[l];
  const RoiGeom g = roi_geometry(roi, lv.spatial_scale[l], P, sampling_ratio, aligned);
  const long long sc = lv.sc[l], sh = lv.sh[l], sw = lv.sw[l];
  float* fbase = reinterpret_cast<float*>(const_cast<void*>(lv.feat[l])) + (long long)bi * lv.sb[l];
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float* f = fbase + (long long)c * sc;
    for (int pw = 0; pw < P; ++pw) {
      const float go = ld_elem(grad_out + r * out_sr + c * out_sc + ph * out_sh + pw * out_sw) / g.count;
      for (int iy = 0; iy < g.gh; ++iy) {
        const float y = g.start_h + (float)ph * g.bin_h + ((float)iy + 0.5f) * g.bin_h / (float)g.gh;
        for (int ix = 0; ix < g.gw; ++ix) {
          const float x = g.start_w + (float)pw * g.bin_w + ((float)ix + 0.5f) * g.bin_w / (float)g.gw;
          const Taps t = bilinear_taps(H, W, y, x);
          if (!t.ok) continue;
          atomicAdd(f + t.yl * sh + t.xl * sw, go * t.w1);
          atomicAdd(f + t.yl * sh + t.xh * sw, go * t.w2);
          atomicAdd(f + t.yh * sh + t.xl * sw, go * t.w3);
          atomicAdd(f + t.yh * sh + t.xh * sw, go * t.w4);
        }
      }
    }
  }
}

// ---- separable fast path (channels-last, sampling_ratio 1 or 2) -----------------------------------
constexpr int kMaxP = 16;  // pooled size the LDS tap tables are sized for

// One axis of bilinear_taps(): the two cells and their weights; a sample outside [-1, size] is flagged
// (the oracle skips it, which is not the same as weight 0 when a feature is non-finite).
struct Tap1 {
  int lo, hi;
  float wlo, whi;
  bool ok;
};
__device__ __forceinline__ Tap1 tap1(int size, float v) {
  Tap1 t;
  t.ok = !(v < -1.0f || v > (float)size);
  if (v <= 0.0f) v = 0.0f;
  int lo = (int)v, hi;
  if (lo >= size - 1) {
    hi = lo = size - 1;
    v = (float)lo;
  } else {
    hi = lo + 1;
  }
  const float l = v - (float)lo;
  t.lo = lo; t.hi = hi;
  t.wlo = 1.0f - l; t.whi = l;
  return t;
}

template <int V>
__device__ __forceinline__ void store_vec(float* p, const float* v) {
#pragma unroll
  for (int i = 0; i < V; i += 4) *reinterpret_cast<float4*>(p + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
}
template <int V>
__device__ __forceinline__ void store_vec(__hip_bfloat16* p, const float* v) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  uint32_t w[V / 2];
#pragma unroll
  for (int i = 0; i < V / 2; ++i) {
    const f32x2 f = {v[2 * i], v[2 * i + 1]};
    const bf16x2 r = __builtin_convertvector(f, bf16x2);
    w[i] = *reinterpret_cast<const uint32_t*>(&r);
  }
  if (V == 8) *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2 % (V / 2)], w[3 % (V / 2)]);
  else *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]);
}

// XCD-aware block -> RoI: grid = 8 * ceil(R/8); blocks with equal blockIdx % 8 share an XCD and get a
// contiguous range of RoIs (RoIs are ordered by image, so an XCD's L2 serves one or two images' maps).
__device__ __forceinline__ int roi_of_block(int R) {
  const int per = (int)(gridDim.x >> 3);
  return (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
}

// Row taps of pooled row ph into (element offset y*sh | -1 = skip, weight / count); duplicates merged.
template <int SR>
__device__ __forceinline__ void row_taps(const RoiGeom& g, int H, int sh, int ph, int* yo, float* yw) {
  const float inv_count = 1.0f / g.count;  // count = SR*SR in {1, 4}: the scaling is exact
#pragma unroll
  for (int iy = 0; iy < SR; ++iy) {
    const float y = g.start_h + (float)ph * g.bin_h + ((float)iy + 0.5f) * g.bin_h / (float)SR;
    const Tap1 t = tap1(H, y);
    yo[2 * iy] = t.ok ? t.lo * sh : -1;
    yo[2 * iy + 1] = t.ok ? t.hi * sh : -1;
    yw[2 * iy] = t.wlo * inv_count;
    yw[2 * iy + 1] = t.whi * inv_count;
  }
  // taps are nondecreasing in y: fold a tap onto an equal earlier one (its own slot is then skipped)
#pragma unroll
  for (int k = 1; k < 2 * SR; ++k) {
#pragma unroll
    for (int m = 0; m < k; ++m) {
      if (yo[k] >= 0 && yo[k] == yo[m]) {
        yw[m] += yw[k];
        yo[k] = -1;
      }
    }
  }
}

template <typename TF, typename TO, int SR>
__global__ __launch_bounds__(512) void roi_align_fwd_cl_kernel(const hd_roi_levels lv, const float* __restrict__ rois,
                                                               int R, int C, int P, int aligned, TO* __restrict__ out,
                                                               long long out_sr, long long out_sh, long long out_sw) {
  constexpr int V = RowVec<TF>::N;
  __shared__ int s_yo[kMaxP][2 * SR];
  __shared__ float s_yw[kMaxP][2 * SR];
  __shared__ int s_xl[kMaxP * SR], s_xh[kMaxP * SR];
  __shared__ float s_xwl[kMaxP * SR], s_xwh[kMaxP * SR];
  const int r = roi_of_block(R);
  if (r >= R) return;  // whole workgroup
  const float* roi = rois + (size_t)r * 5;
  const int l = roi_level_of(roi, lv.k_min, lv.n_levels, lv.canonical_scale, lv.canonical_level);
  const int bi = (int)roi[0];
  const int H = lv.h[l], W = lv.w[l];
  const RoiGeom g = roi_geometry(roi, lv.spatial_scale[l], P, SR, aligned);
  const int sh = (int)lv.sh[l], sw = (int)lv.sw[l];
  for (int i = threadIdx.x; i < 2 * P * SR; i += blockDim.x) {
    if (i < P * SR) {  // column sample i = pw*SR + ix
      const int pw = i / SR, ix = i - pw * SR;
      const float x = g.start_w + (float)pw * g.bin_w + ((float)ix + 0.5f) * g.bin_w / (float)SR;
      const Tap1 t = tap1(W, x);
      s_xl[i] = t.ok ? t.lo * sw : -1;
      s_xh[i] = t.ok ? t.hi * sw : -1;
      s_xwl[i] = t.wlo;
      s_xwh[i] = t.whi;
    } else if ((i - P * SR) % SR == 0) {  // one thread per pooled row builds that row's (merged) taps
      const int ph = (i - P * SR) / SR;
      int yo[2 * SR];
      float yw[2 * SR];
      row_taps<SR>(g, H, sh, ph, yo, yw);
#pragma unroll
      for (int k = 0; k < 2 * SR; ++k) {
        s_yo[ph][k] = yo[k];
        s_yw[ph][k] = yw[k];
      }
    }
  }
  __syncthreads();
  const int cg = C / V;
  const TF* fimg = reinterpret_cast<const TF*>(lv.feat[l]) + (long long)bi * lv.sb[l];
  for (int item = threadIdx.x; item < P * cg; item += blockDim.x) {
    const int ph = item / cg, gq = item - ph * cg;
    const TF* f = fimg + gq * V;
    int yo[2 * SR];
    float yw[2 * SR];
#pragma unroll
    for (int k = 0; k < 2 * SR; ++k) {
      yo[k] = s_yo[ph][k];
      yw[k] = s_yw[ph][k];
    }
    // column value colval(x)[c] = sum_k yw[k] * f[y_k][x][c]: 2*SR independent 16-byte loads
    auto load_col = [&](int xoff, float* cv) {
      uint4 raw[2 * SR];
#pragma unroll
      for (int k = 0; k < 2 * SR; ++k)
        if (yo[k] >= 0) raw[k] = *reinterpret_cast<const uint4*>(f + yo[k] + xoff);
#pragma unroll
      for (int j = 0; j < V; ++j) cv[j] = 0.0f;
#pragma unroll
      for (int k = 0; k < 2 * SR; ++k)
        if (yo[k] >= 0) {
          float v[V];
          RowVec<TF>::unpack(raw[k], v);
#pragma unroll
          for (int j = 0; j < V; ++j) cv[j] = fmaf(yw[k], v[j], cv[j]);
        }
    };
    float c0v[V], c1v[V];  // the two most recent column values, at element offsets c0 <= c1
    int c0 = -1, c1 = -1;
    TO* o = out + (long long)r * out_sr + (long long)ph * out_sh + gq * V;
    for (int pw = 0; pw < P; ++pw) {
      float acc[V];
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] = 0.0f;
#pragma unroll
      for (int ix = 0; ix < SR; ++ix) {
        const int s = pw * SR + ix;
        const int xl = s_xl[s], xh = s_xh[s];  // workgroup-uniform: the branches below do not diverge
        if (xl < 0) continue;
        const float wl = s_xwl[s], wh = s_xwh[s];
        if (xl == c1) {
#pragma unroll
          for (int j = 0; j < V; ++j) c0v[j] = c1v[j];
          c0 = c1;
        } else if (xl != c0) {
          load_col(xl, c0v);
          c0 = xl;
        }
        if (xh == c0) {
#pragma unroll
          for (int j = 0; j < V; ++j) c1v[j] = c0v[j];
          c1 = c0;
        } else if (xh != c1) {
          load_col(xh, c1v);
          c1 = xh;
        }
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = fmaf(wl, c0v[j], fmaf(wh, c1v[j], acc[j]));
      }
      store_vec<V>(o + (long long)pw * out_sw, acc);
    }
  }
}

// grid = R*P workgroups (RoI, ph); thread = channel; f32 gradient maps (channels-last), grad_out f32 | bf16
template <typename TG, int SR>
__global__ __launch_bounds__(256) void roi_align_bwd_cl_kernel(const hd_roi_levels lv, const float* __restrict__ rois,
                                                               int C, int P, int aligned,
                                                               const TG* __restrict__ grad_out, long long out_sr,
                                                               long long out_sh, long long out_sw) {
  __shared__ int s_xl[kMaxP * SR], s_xh[kMaxP * SR];
  __shared__ float s_xwl[kMaxP * SR], s_xwh[kMaxP * SR];
  const int r = (int)(blockIdx.x / (unsigned)P), ph = (int)(blockIdx.x % (unsigned)P);
  const float* roi = rois + (size_t)r * 5;
  const int l = roi_level_of(roi, lv.k_min, lv.n_levels, lv.canonical_scale, lv.canonical_level);
  const int bi = (int)roi[0];
  const int H = lv.h[l], W = lv.w[l];
  const RoiGeom g = roi_geometry(roi, lv.spatial_scale[l], P, SR, aligned);
  const int sh = (int)lv.sh[l], sw = (int)lv.sw[l];
  for (int i = threadIdx.x; i < P * SR; i += blockDim.x) {
    const int pw = i / SR, ix = i - pw * SR;
    const float x = g.start_w + (float)pw * g.bin_w + ((float)ix + 0.5f) * g.bin_w / (float)SR;
    const Tap1 t = tap1(W, x);
    s_xl[i] = t.ok ? t.lo * sw : -1;
    s_xh[i] = t.ok ? t.hi * sw : -1;
    s_xwl[i] = t.wlo;
    s_xwh[i] = t.whi;
  }
  int yo[2 * SR];
  float yw[2 * SR];
  row_taps<SR>(g, H, sh, ph, yo, yw);  // workgroup-uniform (every thread computes the same 2*SR taps)
  __syncthreads();
  float* fimg = reinterpret_cast<float*>(const_cast<void*>(lv.feat[l])) + (long long)bi * lv.sb[l];
  const TG* gbase = grad_out + (long long)r * out_sr + (long long)ph * out_sh;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float* f = fimg + c;
    auto flush = [&](int xoff, float v) {
      if (xoff < 0) return;
#pragma unroll
      for (int k = 0; k < 2 * SR; ++k)
        if (yo[k] >= 0) atomicAdd(f + yo[k] + xoff, yw[k] * v);
    };
    // pending column gradients: column a (value va) and its right neighbour b (value vb)
    int a = -1, b = -1;
    float va = 0.0f, vb = 0.0f;
    float gnext = ld_elem(gbase + c);
    for (int pw = 0; pw < P; ++pw) {
      const float gv = gnext;
      if (pw + 1 < P) gnext = ld_elem(gbase + (long long)(pw + 1) * out_sw + c);
#pragma unroll
      for (int ix = 0; ix < SR; ++ix) {
        const int s = pw * SR + ix;
        const int xl = s_xl[s], xh = s_xh[s];  // workgroup-uniform, nondecreasing in s
        if (xl < 0) continue;
        const float wl = s_xwl[s], wh = s_xwh[s];
        if (xl != a) {
          flush(a, va);
          if (xl == b) {
            a = b;
            va = vb;
          } else {
            flush(b, vb);
            a = xl;
            va = 0.0f;
          }
          b = -1;
          vb = 0.0f;
        }
        va = fmaf(wl, gv, va);
        if (xh == a) {
          va = fmaf(wh, gv, va);
        } else {
          if (xh != b) {
            flush(b, vb);
            b = xh;
            vb = 0.0f;
          }
          vb = fmaf(wh, gv, vb);
        }
      }
    }
    flush(a, va);
    flush(b, vb);
  }
}

// ---- backward as a GATHER: no atomics, every cell written once, bit-reproducible ------------------------------
// (the scatter above runs at the float-atomic rate, 0.60 ms at R = 2048, C = 256, behind a zero-fill of the
// float32 maps and in front of their cast to bf16 — 0.9 + 0.9 + 1.4 GB of traffic for 25 MB of incoming gradient.)
// A WAVE owns an 8 x 8 tile of cells of one (level, image) map and 64 channels (lane = channel); the four waves of a workgroup
// are four channel groups of the same tile and never synchronise.  roi_gather_prep_kernel first writes one 32-byte record per
// RoI (image, level, the geometry the forward uses, the window of cells its samples can touch).  The wave then
//   * scans the records (ballot-compacted in RoI order into a wave-private list: the summation order is fixed);
//   * per listed RoI forms the dense little matrices
//         Wy[cell row][ph] = sum over the SR samples of bin ph of the bilinear weight that sample gives this cell row (x 1/count)
//         Wx[cell col][pw] = the same along x
//     (the forward is out[ph][pw] = sum_y sum_x Wy[y][ph] Wx[x][pw] f[y][x]; this is its transpose), 56 lanes x two passes;
//   * stages its 64 channels of the RoI's incoming gradient block g[7][7][64] through a wave-private LDS buffer with 16-byte
//     loads (the NEXT RoI's chunks travel in registers meanwhile) and accumulates
//         acc[y][x] += sum_pw Wx[x][pw] (sum_ph Wy[y][ph] g[ph][pw])         (840 FMAs per RoI, 64 accumulators per lane);
//   * writes the finished tile through LDS as 16-byte stores in the MAP's dtype: the bf16 maps of a bf16 pyramid are written
//     directly (no float32 accumulator maps, no zero-fill, no cast pass); tiles no RoI touches are written as zeros.
constexpr int kGatherTile = 8;
constexpr int kGatherP = 7;          // pooled size of the instantiation (other sizes take the scatter)
constexpr int kGatherList = 1024;    // wave-private list of RoI indices per scan round
constexpr int kScanUnroll = 8;       // groups of 64 records whose loads are in flight together

struct GatherRec {                   // 32 bytes per RoI
  int img_lvl;                       // image << 8 | level; -1: a RoI no tile takes (image index out of range)
  int ywin, xwin;                    // first | last << 16 cell its samples can touch (conservative), clipped to the map
  float start_h, start_w, bin_h, bin_w;
  int pad;
};

struct GatherPlan {
  int tile_start[HD_MAX_LEVELS + 1];  // first workgroup of each level (x dimension of the grid)
  int tiles_x[HD_MAX_LEVELS], tiles_y[HD_MAX_LEVELS];
  int batch;
};

template <int SR>
__global__ __launch_bounds__(256) void roi_gather_prep_kernel(const hd_roi_levels lv, const float* __restrict__ rois, int R,
                                                              int batch, int aligned, GatherRec* __restrict__ recs,
                                                              int* __restrict__ ranges) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float* roi = rois + (size_t)r * 5;
  const int l = roi_level_of(roi, lv.k_min, lv.n_levels, lv.canonical_scale, lv.canonical_level);
  const int bi = (int)roi[0];
  const RoiGeom g = roi_geometry(roi, lv.spatial_scale[l], kGatherP, SR, aligned);
  GatherRec rec;
  rec.img_lvl = (bi >= 0 && bi < batch) ? ((bi << 8) | l) : -1;
  // cells that can receive weight: floor(first sample) .. floor(last sample) + 1; widened by one more on both sides
  const int H = lv.h[l], W = lv.w[l];
  // aligned = 1 does not clamp the RoI's extent: a malformed RoI (x2 < x1 or y2 < y1) has negative bins, its samples run
  // DOWN from start.  The forward and the scatter backward include them, so the window is the span either way round.
  const float y_e = g.start_h + (float)kGatherP * g.bin_h, x_e = g.start_w + (float)kGatherP * g.bin_w;
  const float y_lo = fminf(g.start_h, y_e), y_hi = fmaxf(g.start_h, y_e);
  const float x_lo = fminf(g.start_w, x_e), x_hi = fmaxf(g.start_w, x_e);
  const int y0 = max(0, min(H - 1, (int)floorf(fmaxf(y_lo, -4.0f)) - 1)), y1 = max(0, min(H - 1, (int)floorf(fminf(y_hi, (float)H + 4.0f)) + 2));
  const int x0 = max(0, min(W - 1, (int)floorf(fmaxf(x_lo, -4.0f)) - 1)), x1 = max(0, min(W - 1, (int)floorf(fminf(x_hi, (float)W + 4.0f)) + 2));
  rec.ywin = y0 | (y1 << 16);
  rec.xwin = x0 | (x1 << 16);
  rec.start_h = g.start_h; rec.start_w = g.start_w; rec.bin_h = g.bin_h; rec.bin_w = g.bin_w;
  rec.pad = 0;
  recs[r] = rec;
  // RoI lists are normally ordered by image: then a tile only has to scan its image's range [first, last] of the list.
  // ranges: [0] = 1 while no descending pair was seen, [1 + b] = first index of image b, [1 + batch + b] = last index + 1
  if (r > 0 && (int)rois[(size_t)(r - 1) * 5] > bi) atomicAnd(&ranges[0], 0);
  if (bi >= 0 && bi < batch) {
    atomicMin(&ranges[1 + bi], r);
    atomicMax(&ranges[1 + batch + bi], r + 1);
  }
}

__global__ void roi_gather_ranges_init_kernel(int* __restrict__ ranges, int batch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) ranges[0] = 1;
  if (i >= 1 && i <= batch) ranges[i] = 0x7fffffff;
  if (i > batch && i <= 2 * batch) ranges[i] = 0;
}

template <typename TG, typename TM>
struct GatherLds {
  static constexpr int P = kGatherP, T = kGatherTile;
  static constexpr int STAGE = P * P * 64 * (int)sizeof(TG);            // one RoI's 64-channel gradient block
  static constexpr int OUT = T * T * 64 * (int)sizeof(TM);
  static constexpr int AREA = (2 * STAGE > OUT ? 2 * STAGE : OUT);
  static constexpr int WBYTES = 2 * 2 * T * 8 * 4;                      // [buf][y | x][cell][8] float
  static constexpr int WAVE = AREA + WBYTES + kGatherList * 4;
  static constexpr int TOTAL = 4 * WAVE;
};

template <typename TG, typename TM, int SR>
__global__ __launch_bounds__(256, 2) void roi_align_bwd_gather_kernel(const hd_roi_levels lv, const GatherPlan plan,
                                                                      const GatherRec* __restrict__ recs,
                                                                      const int* __restrict__ ranges, int R, int C,
                                                                      const TG* __restrict__ grad_out, long long out_sr,
                                                                      long long out_sh, long long out_sw) {
  constexpr int T = kGatherTile, P = kGatherP;
  constexpr int GV = 16 / (int)sizeof(TG), MV = 16 / (int)sizeof(TM);
  constexpr int GCPR = 64 / GV;                       // 16-byte chunks per (ph, pw) row of the wave's 64 channels
  constexpr int NCH = P * P * GCPR, NL = (NCH + 63) / 64;
  using L = GatherLds<TG, TM>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* area = smem + wave * L::WAVE;                        // [2 gradient stages], later the output tile
  float* wbuf = reinterpret_cast<float*>(area + L::AREA);             // [buf][2][T][8]
  int* list = reinterpret_cast<int*>(area + L::AREA + L::WBYTES);
  // which (level, image, tile)
  int l = 0;
  while (l + 1 < lv.n_levels && (int)blockIdx.x >= plan.tile_start[l + 1]) ++l;
  const int local = (int)blockIdx.x - plan.tile_start[l];
  const int per_img = plan.tiles_x[l] * plan.tiles_y[l];
  const int bi = local / per_img, tq = local % per_img;
  const int ty0 = (tq / plan.tiles_x[l]) * T, tx0 = (tq % plan.tiles_x[l]) * T;
  const int H = lv.h[l], W = lv.w[l];
  const int c0 = ((int)blockIdx.y * 4 + wave) * 64;   // this wave's first channel
  if (c0 >= C) return;                                // whole wave (C is a multiple of 64)
  const int key = (bi << 8) | l;

  float acc[T][T];
#pragma unroll
  for (int y = 0; y < T; ++y)
#pragma unroll
    for (int x = 0; x < T; ++x) acc[y][x] = 0.0f;

  // RoI r's [P*P][64 channels] block -> stage `buf`, straight into LDS (global_load_lds_dwordx4: lane L's 16 bytes land at
  // base + 16 L, and the stage is dense in chunk order, so chunk id = lane + 64 k goes to k KiB + 16 lane): no staging registers
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void gvoid;
  auto load_block = [&](int r, int buf) {
    const TG* gb = grad_out + (long long)r * out_sr + c0;
    unsigned char* sb = area + buf * L::STAGE;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int id = lane + 64 * k;
      if (NCH % 64 == 0 || id < NCH) {
        const int p = id / GCPR, q = id % GCPR;
        __builtin_amdgcn_global_load_lds((gvoid*)(gb + (long long)(p / P) * out_sh + (long long)(p % P) * out_sw + q * GV),
                                         (lds_void*)(sb + k * 1024), 16, 0, 0);
      }
    }
  };
  // Wy / Wx of RoI r against this tile: lane = (cell, p); -> row / column masks of the non-zero cells
  unsigned rowmask[2], colmask[2];
  auto weights = [&](int r, int buf) {
    const GatherRec rec = recs[r];
    const int cell = lane >> 3, p = lane & 7;
    float wy = 0.0f, wx = 0.0f;
    if (p < P) {
#pragma unroll
      for (int is = 0; is < SR; ++is) {
        const float vy = rec.start_h + (float)p * rec.bin_h + ((float)is + 0.5f) * rec.bin_h / (float)SR;
        const Tap1 t = tap1(H, vy);
        if (t.ok) {
          if (t.lo == ty0 + cell) wy += t.wlo;
          if (t.hi == ty0 + cell && t.hi != t.lo) wy += t.whi;
        }
        const float vx = rec.start_w + (float)p * rec.bin_w + ((float)is + 0.5f) * rec.bin_w / (float)SR;
        const Tap1 u = tap1(W, vx);
        if (u.ok) {
          if (u.lo == tx0 + cell) wx += u.wlo;
          if (u.hi == tx0 + cell && u.hi != u.lo) wx += u.whi;
        }
      }
      wy *= 1.0f / (float)(SR * SR);   // count = SR * SR in {1, 4}: exact
    }
    wbuf[(buf * 2 + 0) * T * 8 + lane] = wy;
    wbuf[(buf * 2 + 1) * T * 8 + lane] = wx;
    const unsigned long long my = __ballot(wy != 0.0f), mx = __ballot(wx != 0.0f);
    unsigned rm = 0, cm = 0;
#pragma unroll
    for (int c = 0; c < T; ++c) {
      rm |= ((my >> (8 * c)) & 0xffull) ? (1u << c) : 0u;
      cm |= ((mx >> (8 * c)) & 0xffull) ? (1u << c) : 0u;
    }
    rowmask[buf] = rm;
    colmask[buf] = cm;
  };
  auto accumulate = [&](int buf) {
    const unsigned rm = rowmask[buf], cm = colmask[buf];
    if (rm == 0u || cm == 0u) return;
    const TG* sb = reinterpret_cast<const TG*>(area + buf * L::STAGE) + lane;
    const float* wy = wbuf + (buf * 2 + 0) * T * 8;
    const float* wx = wbuf + (buf * 2 + 1) * T * 8;
    float t[T][P];
    {
      float gv[P][P];
#pragma unroll
      for (int ph = 0; ph < P; ++ph)
#pragma unroll
        for (int pw = 0; pw < P; ++pw) gv[ph][pw] = ld_elem(sb + (ph * P + pw) * 64);
#pragma unroll
      for (int y = 0; y < T; ++y) {
#pragma unroll
        for (int pw = 0; pw < P; ++pw) t[y][pw] = 0.0f;
        if (!(rm & (1u << y))) continue;      // wave-uniform
#pragma unroll
        for (int ph = 0; ph < P; ++ph) {
          const float w = wy[y * 8 + ph];
#pragma unroll
          for (int pw = 0; pw < P; ++pw) t[y][pw] = fmaf(w, gv[ph][pw], t[y][pw]);
        }
      }
    }
#pragma unroll
    for (int x = 0; x < T; ++x) {
      if (!(cm & (1u << x))) continue;
      float wxr[P];
#pragma unroll
      for (int pw = 0; pw < P; ++pw) wxr[pw] = wx[x * 8 + pw];
#pragma unroll
      for (int y = 0; y < T; ++y) {
        float a = acc[y][x];
#pragma unroll
        for (int pw = 0; pw < P; ++pw) a = fmaf(wxr[pw], t[y][pw], a);
        acc[y][x] = a;
      }
    }
  };

  // the slice of the RoI list this tile has to look at: its image's range when the list is ordered by image, else all of it
  const bool ordered = ranges[0] != 0;
  int base = ordered ? min(ranges[1 + bi], R) & ~63 : 0;
  R = ordered ? min(ranges[1 + plan.batch + bi], R) : R;
  while (base < R) {
    // ---- scan records until the list is full or the RoIs are exhausted: kScanUnroll groups of 64 records per round, all
    // their loads issued before the first test (one dependent 16-byte L2 load per group made the scan the longest phase:
    // 0.1 us per 64 records per wave, 200 us of a 510 us launch at R = 2048)
    int cnt = 0;
    while (base < R && cnt <= kGatherList - 64 * kScanUnroll) {
      int4 h[kScanUnroll];
#pragma unroll
      for (int u = 0; u < kScanUnroll; ++u) {
        const int r = base + 64 * u + lane;
        h[u] = make_int4(-2, 0, 0, 0);
        if (r < R) h[u] = *reinterpret_cast<const int4*>(recs + r);   // img_lvl, ywin, xwin, start_h (unused here)
      }
#pragma unroll
      for (int u = 0; u < kScanUnroll; ++u) {
        const bool hit = h[u].x == key && (h[u].y & 0xffff) <= ty0 + T - 1 && (h[u].y >> 16) >= ty0 &&
                         (h[u].z & 0xffff) <= tx0 + T - 1 && (h[u].z >> 16) >= tx0;
        const unsigned long long m = __ballot(hit);
        if (hit) list[cnt + __popcll(m & ((1ull << lane) - 1ull))] = base + 64 * u + lane;
        cnt += __popcll(m);
      }
      base += 64 * kScanUnroll;
    }
    if (cnt == 0) continue;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the list, before the wave reads it back
    // ---- accumulate: block i in stage i & 1 while block i + 1 lands in the other stage
    int r0 = __builtin_amdgcn_readfirstlane(list[0]);
    load_block(r0, 0);
    weights(r0, 0);
    for (int i = 0; i < cnt; ++i) {
      const int buf = i & 1;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // block i has landed, its weights are written
      int rn = 0;
      if (i + 1 < cnt) {
        rn = __builtin_amdgcn_readfirstlane(list[i + 1]);
        if (buf == 0) load_block(rn, 1); else load_block(rn, 0);     // stage buf ^ 1 was last read for block i - 1
      }
      if (buf == 0) accumulate(0); else accumulate(1);
      if (i + 1 < cnt) {
        if (buf == 0) weights(rn, 1); else weights(rn, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }

  // ---- the tile, through LDS, as 16-byte stores in the map's dtype: out[cell][64 channels]
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  TM* so = reinterpret_cast<TM*>(area);
#pragma unroll
  for (int y = 0; y < T; ++y)
#pragma unroll
    for (int x = 0; x < T; ++x) st_elem(so + (y * T + x) * 64 + lane, acc[y][x]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  TM* fimg = reinterpret_cast<TM*>(const_cast<void*>(lv.feat[l])) + (long long)bi * lv.sb[l] + c0;
  constexpr int OCPR = 64 / MV;   // 16-byte chunks per cell
#pragma unroll
  for (int k = 0; k < T * T * OCPR / 64; ++k) {
    const int id = lane + 64 * k;
    const int cell = id / OCPR, q = id % OCPR;
    const int y = ty0 + cell / T, x = tx0 + cell % T;
    if (y < H && x < W)
      *reinterpret_cast<uint4*>(fimg + (long long)y * lv.sh[l] + (long long)x * lv.sw[l] + q * MV) =
          *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(so) + id * 16);
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
  const float* fbase = reinterpret_cast<const float*>(lv.feat[l]) + (long long)bi * lv.sb[l];
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
  float* fbase = reinterpret_cast<float*>(const_cast<void*>(lv.feat[l])) + (long long)bi * lv.sb[l];
  for (int c = threadIdx.x; c < C; c += blockDim.x)
    for (int pw = 0; pw < P; ++pw) {
      const long long o = r * out_sr + c * out_sc + ph * out_sh + pw * out_sw;
      const int a = argmax[o];
      if (a >= 0) atomicAdd(fbase + (long long)c * sc + (long long)(a / W) * sh + (long long)(a % W) * sw, grad_out[o]);
    }
}

inline hd_status check_roi_args(const hd_roi_levels* lv, const float* rois, int R, int C, int P, const void* out) {
  if (!lv) return HD_ERR_NULL;
  if (R < 0 || C <= 0 || P <= 0 || P > 65535 || lv->n_levels <= 0 || lv->n_levels > HD_MAX_LEVELS) return HD_ERR_SHAPE;
  if (lv->dtype != 0 && lv->dtype != 1) return HD_ERR_SHAPE;
  if (R == 0) return HD_OK;
  if (!rois || !out) return HD_ERR_NULL;
  for (int l = 0; l < lv->n_levels; ++l) {
    if (!lv->feat[l]) return HD_ERR_NULL;
    if (lv->h[l] <= 0 || lv->w[l] <= 0) return HD_ERR_SHAPE;
  }
  return HD_OK;
}

// The separable fast path needs channels-last maps whose 16-byte vectors are aligned, 32-bit in-image offsets,
// and a channels-last output/gradient tensor (out_sc == 1) aligned for its vector stores.
inline bool roi_fast_ok(const hd_roi_levels* lv, int C, int P, int sampling_ratio, int feat_elem, int vec,
                        const void* out, int out_elem, int out_vec, long long out_sr, long long out_sc, long long out_sh,
                        long long out_sw) {
  if (sampling_ratio != 1 && sampling_ratio != 2) return false;
  if (P > kMaxP || C % vec != 0 || out_sc != 1) return false;
  const long long fa = 16 / feat_elem;  // elements per 16 bytes
  for (int l = 0; l < lv->n_levels; ++l) {
    if (lv->sc[l] != 1 || !hd_aligned16(lv->feat[l])) return false;
    if (lv->sb[l] % fa || lv->sh[l] % fa || lv->sw[l] % fa) return false;
    if (lv->sh[l] <= 0 || lv->sw[l] <= 0 || (long long)lv->h[l] * lv->sh[l] >= (1ll << 31) ||
        (long long)lv->w[l] * lv->sw[l] >= (1ll << 31))
      return false;
  }
  const long long oa = out_vec;  // output vectors of out_vec elements must stay aligned
  if ((reinterpret_cast<uintptr_t>(out) % (size_t)(out_vec * out_elem > 16 ? 16 : out_vec * out_elem)) != 0) return false;
  if (out_sr % oa || out_sh % oa || out_sw % oa) return false;
  return true;
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
                                      int sampling_ratio, int aligned, void* out, int out_dtype, int64_t out_sr,
                                      int64_t out_sc, int64_t out_sh, int64_t out_sw, hd_stream stream) {
  hd_status st = check_roi_args(levels, rois, R, C, P, out);
  if (st != HD_OK || R == 0) return st;
  if (out_dtype != 0 && out_dtype != 1) return HD_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const bool fbf = levels->dtype == 1, obf = out_dtype == 1;
  const int vec = fbf ? 8 : 4;
  if (roi_fast_ok(levels, C, P, sampling_ratio, fbf ? 2 : 4, vec, out, obf ? 2 : 4, vec, out_sr, out_sc, out_sh, out_sw)) {
    int threads = (P * (C / vec) + 63) / 64 * 64;
    threads = threads < 64 ? 64 : (threads > 512 ? 512 : threads);
    const unsigned grid = (unsigned)((R + 7) / 8) * 8u;
#define HD_ROI_FWD(TF, TO, SR)                                                                                       \
  (void)hd_launch(roi_align_fwd_cl_kernel<TF, TO, SR>, dim3(grid), dim3(threads), 0, s, *levels, rois, R, C, P, \
                     aligned, reinterpret_cast<TO*>(out), (long long)out_sr, (long long)out_sh, (long long)out_sw)
#define HD_ROI_FWD_SR(TF, TO)              \
  do {                                     \
    if (sampling_ratio == 1) HD_ROI_FWD(TF, TO, 1); \
    else HD_ROI_FWD(TF, TO, 2);            \
  } while (0)
    if (fbf && obf) HD_ROI_FWD_SR(__hip_bfloat16, __hip_bfloat16);
    else if (fbf) HD_ROI_FWD_SR(__hip_bfloat16, float);
    else if (obf) HD_ROI_FWD_SR(float, __hip_bfloat16);
    else HD_ROI_FWD_SR(float, float);
#undef HD_ROI_FWD_SR
#undef HD_ROI_FWD
    HD_RETURN_IF_LAUNCH_FAILED();
    return HD_OK;
  }
  const int threads = C > 128 ? 256 : (C > 64 ? 128 : 64);
#define HD_ROI_GEN(TF, TO)                                                                                          \
  hipLaunchKernelGGL((roi_align_fwd_kernel<TF, TO>), dim3(R, P), dim3(threads), 0, s, *levels, rois, C, P,          \
                     sampling_ratio, aligned, reinterpret_cast<TO*>(out), (long long)out_sr, (long long)out_sc,     \
                     (long long)out_sh, (long long)out_sw)
  if (fbf && obf) HD_ROI_GEN(__hip_bfloat16, __hip_bfloat16);
  else if (fbf) HD_ROI_GEN(__hip_bfloat16, float);
  else if (obf) HD_ROI_GEN(float, __hip_bfloat16);
  else HD_ROI_GEN(float, float);
#undef HD_ROI_GEN
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_roi_align_bwd(const hd_roi_levels* grad_levels, const float* rois, int R, int C, int P,
                                      int sampling_ratio, int aligned, const void* grad_out, int grad_dtype,
                                      int64_t out_sr, int64_t out_sc, int64_t out_sh, int64_t out_sw,
                                      hd_stream stream) {
  hd_status st = check_roi_args(grad_levels, rois, R, C, P, grad_out);
  if (st != HD_OK || R == 0) return st;
  if (grad_levels->dtype != 0) return HD_ERR_SHAPE;  // the accumulators are float32
  if (grad_dtype != 0 && grad_dtype != 1) return HD_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const bool gbf = grad_dtype == 1;
  // lane = channel: only the maps' channel stride and 32-bit offsets matter (vec = 1, any alignment)
  if (roi_fast_ok(grad_levels, C, P, sampling_ratio, 16, 1, grad_out, 1, 1, out_sr, out_sc, out_sh, out_sw) &&
      (long long)R * P < (1ll << 31)) {
    const int threads = C > 128 ? 256 : (C > 64 ? 128 : 64);
    const unsigned grid = (unsigned)((long long)R * P);
#define HD_ROI_BWD(TG, SR)                                                                                     \
  (void)hd_launch(roi_align_bwd_cl_kernel<TG, SR>, dim3(grid), dim3(threads), 0, s, *grad_levels, rois, C, P, \
                     aligned, reinterpret_cast<const TG*>(grad_out), (long long)out_sr, (long long)out_sh,     \
                     (long long)out_sw)
    if (gbf) {
      if (sampling_ratio == 1) HD_ROI_BWD(__hip_bfloat16, 1);
      else HD_ROI_BWD(__hip_bfloat16, 2);
    } else {
      if (sampling_ratio == 1) HD_ROI_BWD(float, 1);
      else HD_ROI_BWD(float, 2);
    }
#undef HD_ROI_BWD
    HD_RETURN_IF_LAUNCH_FAILED();
    return HD_OK;
  }
  const int threads = C > 128 ? 256 : (C > 64 ? 128 : 64);
  if (gbf)
    hipLaunchKernelGGL((roi_align_bwd_kernel<__hip_bfloat16>), dim3(R, P), dim3(threads), 0, s, *grad_levels, rois, C, P,
                       sampling_ratio, aligned, reinterpret_cast<const __hip_bfloat16*>(grad_out), (long long)out_sr,
                       (long long)out_sc, (long long)out_sh, (long long)out_sw);
  else
    hipLaunchKernelGGL((roi_align_bwd_kernel<float>), dim3(R, P), dim3(threads), 0, s, *grad_levels, rois, C, P,
                       sampling_ratio, aligned, reinterpret_cast<const float*>(grad_out), (long long)out_sr,
                       (long long)out_sc, (long long)out_sh, (long long)out_sw);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" int hd_roi_align_bwd_gather_supported(const hd_roi_levels* grad_levels, int R, int C, int P, int sampling_ratio,
                                                 int grad_dtype, int64_t out_sc) {
  if (!grad_levels || R < 0 || C < 64 || C % 64 || P != kGatherP) return 0;
  if (sampling_ratio != 1 && sampling_ratio != 2) return 0;
  if (grad_dtype != 0 && grad_dtype != 1) return 0;
  if (grad_levels->dtype != 0 && grad_levels->dtype != 1) return 0;
  if (out_sc != 1) return 0;
  const long long ma = grad_levels->dtype == 0 ? 4 : 8;  // map elements per 16 bytes
  for (int l = 0; l < grad_levels->n_levels; ++l) {
    if (grad_levels->sc[l] != 1 || !hd_aligned16(grad_levels->feat[l])) return 0;
    if (grad_levels->sb[l] % ma || grad_levels->sh[l] % ma || grad_levels->sw[l] % ma) return 0;
    if (grad_levels->h[l] < 1 || grad_levels->w[l] < 1 || grad_levels->h[l] > 32767 || grad_levels->w[l] > 32767) return 0;
  }
  return 1;
}

static size_t gather_ranges_bytes(int batch) { return ((size_t)(1 + 2 * (batch > 0 ? batch : 1)) * sizeof(int) + 31) / 32 * 32; }
extern "C" size_t hd_roi_align_bwd_gather_workspace_bytes(int R, int batch) {
  return gather_ranges_bytes(batch) + (size_t)(R > 0 ? R : 1) * sizeof(GatherRec);
}

extern "C" hd_status hd_roi_align_bwd_gather(const hd_roi_levels* grad_levels, int batch, const float* rois, int R, int C, int P,
                                             int sampling_ratio, int aligned, const void* grad_out, int grad_dtype,
                                             int64_t out_sr, int64_t out_sc, int64_t out_sh, int64_t out_sw, void* workspace,
                                             size_t workspace_bytes, hd_stream stream) {
  if (!grad_levels || !workspace || (R > 0 && (!rois || !grad_out))) return HD_ERR_NULL;
  if (batch < 1 || batch >= (1 << 22) || grad_levels->n_levels < 1 || grad_levels->n_levels > HD_MAX_LEVELS) return HD_ERR_SHAPE;
  if (!hd_roi_align_bwd_gather_supported(grad_levels, R, C, P, sampling_ratio, grad_dtype, out_sc)) return HD_ERR_SHAPE;
  if (workspace_bytes < hd_roi_align_bwd_gather_workspace_bytes(R, batch)) return HD_ERR_WORKSPACE;
  const int gv = grad_dtype == 0 ? 4 : 8;
  if (!hd_aligned16(workspace) || (R > 0 && (!hd_aligned16(grad_out) || out_sr % gv || out_sh % gv || out_sw % gv))) return HD_ERR_ALIGN;
  GatherPlan plan{};
  int total = 0;
  for (int l = 0; l < grad_levels->n_levels; ++l) {
    plan.tile_start[l] = total;
    plan.tiles_x[l] = (grad_levels->w[l] + kGatherTile - 1) / kGatherTile;
    plan.tiles_y[l] = (grad_levels->h[l] + kGatherTile - 1) / kGatherTile;
    total += plan.tiles_x[l] * plan.tiles_y[l] * batch;
  }
  for (int l = grad_levels->n_levels; l <= HD_MAX_LEVELS; ++l) plan.tile_start[l] = total;
  plan.batch = batch;
  hipStream_t s = (hipStream_t)stream;
  int* ranges = static_cast<int*>(workspace);
  GatherRec* recs = reinterpret_cast<GatherRec*>(static_cast<unsigned char*>(workspace) + gather_ranges_bytes(batch));
  const HdLaunchTiming armed = hd_launch_timing();   // hd_profile_time_next_launch stamps the gather kernel, not the two
  hd_launch_timing() = HdLaunchTiming{};             // preparation launches in front of it
  (void)hd_launch(roi_gather_ranges_init_kernel, dim3((unsigned)((2 * batch + 256) / 256)), dim3(256), 0, s, ranges, batch);
  HD_RETURN_IF_LAUNCH_FAILED();
  if (R > 0) {
    const unsigned pg = (unsigned)((R + 255) / 256);
    if (sampling_ratio == 1)
      (void)hd_launch(roi_gather_prep_kernel<1>, dim3(pg), dim3(256), 0, s, *grad_levels, rois, R, batch, aligned, recs, ranges);
    else
      (void)hd_launch(roi_gather_prep_kernel<2>, dim3(pg), dim3(256), 0, s, *grad_levels, rois, R, batch, aligned, recs, ranges);
    HD_RETURN_IF_LAUNCH_FAILED();
  }
  hd_launch_timing() = armed;
  const dim3 grid((unsigned)total, (unsigned)((C + 255) / 256));
#define HD_ROI_GATHER(TG, TM, SR)                                                                                          \
  (void)hd_launch(roi_align_bwd_gather_kernel<TG, TM, SR>, grid, dim3(256), (size_t)GatherLds<TG, TM>::TOTAL, s, *grad_levels, \
                  plan, (const GatherRec*)recs, (const int*)ranges, R, C, reinterpret_cast<const TG*>(grad_out), (long long)out_sr, \
                  (long long)out_sh, (long long)out_sw)
  const bool gbf = grad_dtype == 1, mbf = grad_levels->dtype == 1;
  if (sampling_ratio == 1) {
    if (gbf && mbf) HD_ROI_GATHER(__hip_bfloat16, __hip_bfloat16, 1);
    else if (gbf) HD_ROI_GATHER(__hip_bfloat16, float, 1);
    else if (mbf) HD_ROI_GATHER(float, __hip_bfloat16, 1);
    else HD_ROI_GATHER(float, float, 1);
  } else {
    if (gbf && mbf) HD_ROI_GATHER(__hip_bfloat16, __hip_bfloat16, 2);
    else if (gbf) HD_ROI_GATHER(__hip_bfloat16, float, 2);
    else if (mbf) HD_ROI_GATHER(float, __hip_bfloat16, 2);
    else HD_ROI_GATHER(float, float, 2);
  }
#undef HD_ROI_GATHER
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_roi_pool_fwd(const hd_roi_levels* levels, const float* rois, int R, int C, int P, float* out,
                                     int32_t* argmax, int64_t out_sr, int64_t out_sc, int64_t out_sh,
                                     int64_t out_sw, hd_stream stream) {
  hd_status st = check_roi_args(levels, rois, R, C, P, out);
  if (st != HD_OK || R == 0) return st;
  if (levels->dtype != 0) return HD_ERR_SHAPE;  // the pooling variant is float32-only
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
  if (grad_levels->dtype != 0) return HD_ERR_SHAPE;
  if (!argmax) return HD_ERR_NULL;
  const int threads = C > 128 ? 256 : (C > 64 ? 128 : 64);
  hipLaunchKernelGGL(roi_pool_bwd_kernel, dim3(R, P), dim3(threads), 0, (hipStream_t)stream, *grad_levels, rois, C, P,
                     grad_out, argmax, (long long)out_sr, (long long)out_sc, (long long)out_sh, (long long)out_sw);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}
