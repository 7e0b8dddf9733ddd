// pool.hip — max-pools on channels-last tensors, forward + backward: 5x5 / stride 1 / pad 2 (the SPPF block of the
// YOLOv5 backbone, reference README.md:9) and 3x3 / stride 2 / pad 1 (the ResNet50 stem, README.md:8).
//
// Why hand-written: PyTorch-ROCm's max_pool_backward_nhwc takes ~325 us per call on a
// [16,256,20,20] bf16 tensor (atomic scatter); SPPF calls it three times per step (1.0 ms of a
// 12 ms step).  Here the forward records the winning tap (0..24, one byte per element) and the
// backward is the GATHER form — every input element sums the gradients of the <= 25 outputs whose
// recorded tap points at it — so there are no atomics and the result is deterministic.
// Lanes run along channels (8 bf16 or 4 f32 = 16 bytes per lane), so every access is a coalesced
// 16-byte vector; the 25-tap reuse is served by L1/L2 (the whole tensor is a few MB).
// Ties: first tap in (dy, dx) scan order wins (strict >), NaN propagates like PyTorch.
#include <hip/hip_bf16.h>

#include "hd_common.h"

namespace {

template <typename T>
struct Vec;
template <>
struct Vec<float> {
  static constexpr int N = 4;
  using Raw = float4;
  static __device__ __forceinline__ void unpack(const uint4 r, float* v) {   // the 16 bytes of one load -> N floats
    v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y); v[2] = __uint_as_float(r.z); v[3] = __uint_as_float(r.w);
  }
  static __device__ __forceinline__ void load(const float* p, float* v) {
    const float4 r = *reinterpret_cast<const float4*>(p);
    v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
  }
  static __device__ __forceinline__ void store(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <>
struct Vec<__hip_bfloat16> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void unpack(const uint4 r, float* v) {   // the 16 bytes of one load -> N floats
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void load(const __hip_bfloat16* p, float* v) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store(__hip_bfloat16* p, const float* v) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // values are exact bf16 (max of bf16 inputs) or sums rounded to nearest-even below
      const __hip_bfloat16 lo = __float2bfloat16(v[2 * i]), hi = __float2bfloat16(v[2 * i + 1]);
      w[i] = (uint32_t)(*reinterpret_cast<const uint16_t*>(&lo)) | ((uint32_t)(*reinterpret_cast<const uint16_t*>(&hi)) << 16);
    }
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

template <typename T>
__global__ __launch_bounds__(256) void maxpool5_fwd_kernel(const T* __restrict__ in, long long ld_in, int B, int H, int W,
                                                           int C, T* __restrict__ out, long long ld_out,
                                                           uint8_t* __restrict__ idx) {
  // ld_in / ld_out: elements between consecutive pixels of in / out (C = dense; > C = a channel slice of a wider
  // channels-last tensor, e.g. a slot of SPPF's concatenation buffer); idx is always dense
  constexpr int V = Vec<T>::N;
  const int cv = C / V;
  const uint32_t total = (uint32_t)B * H * W * cv;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const uint32_t c0 = (i % cv) * V;
    uint32_t r = i / cv;
    const int x = r % W;
    r /= W;
    const int y = r % H;
    const int b = r / H;
    float best[V];
    uint32_t bi[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      best[k] = -INFINITY;
      bi[k] = 12;  // centre tap if every value is -inf
    }
    for (int ty = 0; ty < 5; ++ty) {
      const int yy = y + ty - 2;
      if (yy < 0 || yy >= H) continue;
      // the five taps of a row are requested together (clamped columns; out-of-range ones are skipped below): a
      // load -> compare chain per tap made the kernel 25 dependent L2 round trips long
      float v[5][V];
#pragma unroll
      for (int tx = 0; tx < 5; ++tx) {
        const int xx = min(max(x + tx - 2, 0), W - 1);
        Vec<T>::load(in + (((size_t)b * H + yy) * W + xx) * ld_in + c0, v[tx]);
      }
#pragma unroll
      for (int tx = 0; tx < 5; ++tx) {
        const int xx = x + tx - 2;
        if (xx < 0 || xx >= W) continue;
#pragma unroll
        for (int k = 0; k < V; ++k)
          if (v[tx][k] > best[k] || v[tx][k] != v[tx][k]) {
            best[k] = v[tx][k];
            bi[k] = ty * 5 + tx;
          }
      }
    }
    const size_t o = (((size_t)b * H + y) * W + x) * C + c0;
    Vec<T>::store(out + (((size_t)b * H + y) * W + x) * ld_out + c0, best);
    if (V == 8) {
      *reinterpret_cast<uint2*>(idx + o) = make_uint2(bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24),
                                                      bi[4 % V] | (bi[5 % V] << 8) | (bi[6 % V] << 16) | (bi[7 % V] << 24));
    } else {
      *reinterpret_cast<uint32_t*>(idx + o) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool5_bwd_kernel(const T* __restrict__ gout, long long ld_gout,
                                                           const uint8_t* __restrict__ idx, int B, int H, int W, int C,
                                                           T* __restrict__ gin) {
  constexpr int V = Vec<T>::N;
  const int cv = C / V;
  const uint32_t total = (uint32_t)B * H * W * cv;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const uint32_t c0 = (i % cv) * V;
    uint32_t r = i / cv;
    const int x = r % W;
    r /= W;
    const int y = r % H;
    const int b = r / H;
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.0f;
    // output q = p - (t - 2) used tap t on input p
    for (int ty = 0; ty < 5; ++ty) {
      const int qy = y - ty + 2;
      if (qy < 0 || qy >= H) continue;
      // the tap bytes of the five windows of this row are requested together (clamped columns); the gradient of a window is
      // loaded only when one of its taps points here
      uint2 tb[5];
#pragma unroll
      for (int tx = 0; tx < 5; ++tx) {
        const int qx = min(max(x - tx + 2, 0), W - 1);
        const size_t q = (((size_t)b * H + qy) * W + qx) * C + c0;
        if (V == 8) tb[tx] = *reinterpret_cast<const uint2*>(idx + q);
        else tb[tx] = make_uint2(*reinterpret_cast<const uint32_t*>(idx + q), 0u);
      }
#pragma unroll
      for (int tx = 0; tx < 5; ++tx) {
        const int qx = x - tx + 2;
        if (qx < 0 || qx >= W) continue;
        uint32_t taps[V];
#pragma unroll
        for (int k = 0; k < V; ++k) taps[k] = ((k < 4 ? tb[tx].x : tb[tx].y) >> (8 * (k & 3))) & 255u;
        const uint32_t want = ty * 5 + tx;
        bool any = false;
#pragma unroll
        for (int k = 0; k < V; ++k) any |= (taps[k] == want);
        if (!any) continue;
        float g[V];
        Vec<T>::load(gout + (((size_t)b * H + qy) * W + qx) * ld_gout + c0, g);
#pragma unroll
        for (int k = 0; k < V; ++k)
          if (taps[k] == want) acc[k] += g[k];
      }
    }
    Vec<T>::store(gin + (((size_t)b * H + y) * W + x) * C + c0, acc);
  }
}

// ---- 3x3 / stride 2 / pad 1 (the ResNet stem's pooling, reference README.md:8 FasterRCNN-R50) ---------------------------
// Same scheme: the forward records the winning tap (0..8, one byte per OUTPUT element), the backward is the gather form —
// an input pixel is covered by at most 2 x 2 output windows — so there is no int64 index tensor (PyTorch writes and reads
// 8 bytes per output element) and no atomic scatter: 0.29 ms -> ~0.1 ms per backward call at [16, 64, 416, 416] bf16.
template <typename T>
__global__ __launch_bounds__(256) void maxpool3s2_fwd_kernel(const T* __restrict__ in, int B, int H, int W, int C, int Ho,
                                                             int Wo, T* __restrict__ out, uint8_t* __restrict__ idx) {
  constexpr int V = Vec<T>::N;
  const int cv = C / V;
  const uint32_t total = (uint32_t)B * Ho * Wo * cv;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const uint32_t c0 = (i % cv) * V;
    uint32_t r = i / cv;
    const int ox = r % Wo;
    r /= Wo;
    const int oy = r % Ho;
    const int b = r / Ho;
    float best[V];
    uint32_t bi[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      best[k] = -INFINITY;
      bi[k] = 4;  // centre tap if every value is -inf
    }
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      const int yy = 2 * oy + ty - 1;
      if (yy < 0 || yy >= H) continue;
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const int xx = 2 * ox + tx - 1;
        if (xx < 0 || xx >= W) continue;
        float v[V];
        Vec<T>::load(in + (((size_t)b * H + yy) * W + xx) * C + c0, v);
#pragma unroll
        for (int k = 0; k < V; ++k)
          if (v[k] > best[k] || v[k] != v[k]) {
            best[k] = v[k];
            bi[k] = ty * 3 + tx;
          }
      }
    }
    const size_t o = (((size_t)b * Ho + oy) * Wo + ox) * C + c0;
    Vec<T>::store(out + o, best);
    if (V == 8) {
      *reinterpret_cast<uint2*>(idx + o) = make_uint2(bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24),
                                                      bi[4 % V] | (bi[5 % V] << 8) | (bi[6 % V] << 16) | (bi[7 % V] << 24));
    } else {
      *reinterpret_cast<uint32_t*>(idx + o) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const T* __restrict__ gout, const uint8_t* __restrict__ idx, int B,
                                                             int H, int W, int C, int Ho, int Wo, T* __restrict__ gin) {
  // A thread owns a 2 x 2 block of input pixels (rows 2j, 2j+1; columns 2i, 2i+1) of one channel group.  The block sits under
  // the four windows (j | j+1, i | i+1): their tap bytes and gradients are loaded once (96 bytes) for 64 bytes of output,
  // all requested before the first comparison.  (One thread per input pixel re-read every window four times and was bound by
  // that L2 traffic: 264 -> 192 us with batched loads, this form is the next step.)
  //   (2j, 2i)      <- window (j, i) tap (1,1)
  //   (2j, 2i+1)    <- (j, i+1) tap (1,0), (j, i) tap (1,2)
  //   (2j+1, 2i)    <- (j+1, i) tap (0,1), (j, i) tap (2,1)
  //   (2j+1, 2i+1)  <- (j+1, i+1) tap (0,0), (j+1, i) tap (0,2), (j, i+1) tap (2,0), (j, i) tap (2,2)
  constexpr int V = Vec<T>::N;
  const int cv = C / V;
  const int Hb = (H + 1) / 2, Wb = (W + 1) / 2;
  const uint32_t total = (uint32_t)B * Hb * Wb * cv;
  for (uint32_t it = blockIdx.x * blockDim.x + threadIdx.x; it < total; it += gridDim.x * blockDim.x) {
    const uint32_t c0 = (it % cv) * V;
    uint32_t r = it / cv;
    const int i = r % Wb;
    r /= Wb;
    const int j = r % Hb;
    const int b = r / Hb;
    uint2 tb[4];
    float g[4][V];
    bool ok[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {  // w = 2 * (row offset) + (column offset)
      const int qy = j + (w >> 1), qx = i + (w & 1);
      ok[w] = qy < Ho && qx < Wo;
      const size_t q = ok[w] ? (((size_t)b * Ho + qy) * Wo + qx) * C + c0 : 0;
      if (V == 8) tb[w] = *reinterpret_cast<const uint2*>(idx + q);
      else tb[w] = make_uint2(*reinterpret_cast<const uint32_t*>(idx + q), 0u);
      Vec<T>::load(gout + q, g[w]);
    }
    float acc[4][V];  // [2 * dy + dx]
#pragma unroll
    for (int k = 0; k < V; ++k) {
      uint32_t tap[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) tap[w] = ok[w] ? (((k < 4 ? tb[w].x : tb[w].y) >> (8 * (k & 3))) & 255u) : 255u;
      acc[0][k] = tap[0] == 4 ? g[0][k] : 0.0f;
      acc[1][k] = (tap[1] == 3 ? g[1][k] : 0.0f) + (tap[0] == 5 ? g[0][k] : 0.0f);
      acc[2][k] = (tap[2] == 1 ? g[2][k] : 0.0f) + (tap[0] == 7 ? g[0][k] : 0.0f);
      acc[3][k] = ((tap[3] == 0 ? g[3][k] : 0.0f) + (tap[2] == 2 ? g[2][k] : 0.0f)) +
                  ((tap[1] == 6 ? g[1][k] : 0.0f) + (tap[0] == 8 ? g[0][k] : 0.0f));
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int y = 2 * j + (p >> 1), x = 2 * i + (p & 1);
      if (y < H && x < W) Vec<T>::store(gin + (((size_t)b * H + y) * W + x) * C + c0, acc[p]);
    }
  }
}

inline unsigned pool_grid(long long items) {
  long long b = (items + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

inline hd_status pool_checks(const void* a, const void* b, const void* c, int dtype, int B, int H, int W, int C) {
  if (!a || !b || !c) return HD_ERR_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (dtype != 0 && dtype != 1)) return HD_ERR_SHAPE;
  const int V = dtype == 0 ? 4 : 8;
  if (C % V != 0 || (long long)B * H * W * (C / V) > 0xffffffffLL) return HD_ERR_SHAPE;
  if (!hd_aligned16(a) || !hd_aligned16(b) || (reinterpret_cast<uintptr_t>(c) & 7u)) return HD_ERR_ALIGN;
  return HD_OK;
}

}  // namespace

static hd_status pool_stride(int64_t given, int C, int V, int64_t* ld) {
  *ld = given == 0 ? C : given;
  return (*ld < C || *ld % V != 0) ? HD_ERR_SHAPE : HD_OK;
}

extern "C" hd_status hd_maxpool5_fwd(const void* in, int64_t in_row_stride, int dtype, int B, int H, int W, int C,
                                     void* out, int64_t out_row_stride, uint8_t* idx, hd_stream stream) {
  hd_status st = pool_checks(in, out, idx, dtype, B, H, W, C);
  if (st != HD_OK) return st;
  int64_t ld_in = 0, ld_out = 0;
  if (pool_stride(in_row_stride, C, dtype == 0 ? 4 : 8, &ld_in) != HD_OK ||
      pool_stride(out_row_stride, C, dtype == 0 ? 4 : 8, &ld_out) != HD_OK)
    return HD_ERR_SHAPE;
  const long long items = (long long)B * H * W * (C / (dtype == 0 ? 4 : 8));
  if (dtype == 0)
    hipLaunchKernelGGL((maxpool5_fwd_kernel<float>), dim3(pool_grid(items)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float*>(in), (long long)ld_in, B, H, W, C, reinterpret_cast<float*>(out),
                       (long long)ld_out, idx);
  else
    hipLaunchKernelGGL((maxpool5_fwd_kernel<__hip_bfloat16>), dim3(pool_grid(items)), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<const __hip_bfloat16*>(in), (long long)ld_in, B, H, W, C,
                       reinterpret_cast<__hip_bfloat16*>(out), (long long)ld_out, idx);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_maxpool5_bwd(const void* grad_out, int64_t grad_out_row_stride, const uint8_t* idx, int dtype,
                                     int B, int H, int W, int C, void* grad_in, hd_stream stream) {
  hd_status st = pool_checks(grad_out, grad_in, idx, dtype, B, H, W, C);
  if (st != HD_OK) return st;
  int64_t ld_g = 0;
  if (pool_stride(grad_out_row_stride, C, dtype == 0 ? 4 : 8, &ld_g) != HD_OK) return HD_ERR_SHAPE;
  const long long items = (long long)B * H * W * (C / (dtype == 0 ? 4 : 8));
  if (dtype == 0)
    hipLaunchKernelGGL((maxpool5_bwd_kernel<float>), dim3(pool_grid(items)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float*>(grad_out), (long long)ld_g, idx, B, H, W, C,
                       reinterpret_cast<float*>(grad_in));
  else
    hipLaunchKernelGGL((maxpool5_bwd_kernel<__hip_bfloat16>), dim3(pool_grid(items)), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<const __hip_bfloat16*>(grad_out), (long long)ld_g, idx, B, H, W,
                       C, reinterpret_cast<__hip_bfloat16*>(grad_in));
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_maxpool3s2_fwd(const void* in, int dtype, int B, int H, int W, int C, void* out, uint8_t* idx,
                                       hd_stream stream) {
  hd_status st = pool_checks(in, out, idx, dtype, B, H, W, C);
  if (st != HD_OK) return st;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long items = (long long)B * Ho * Wo * (C / (dtype == 0 ? 4 : 8));
  if (dtype == 0)
    hipLaunchKernelGGL((maxpool3s2_fwd_kernel<float>), dim3(pool_grid(items)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float*>(in), B, H, W, C, Ho, Wo, reinterpret_cast<float*>(out), idx);
  else
    hipLaunchKernelGGL((maxpool3s2_fwd_kernel<__hip_bfloat16>), dim3(pool_grid(items)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const __hip_bfloat16*>(in), B, H, W, C, Ho, Wo, reinterpret_cast<__hip_bfloat16*>(out),
                       idx);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_maxpool3s2_bwd(const void* grad_out, const uint8_t* idx, int dtype, int B, int H, int W, int C,
                                       void* grad_in, hd_stream stream) {
  hd_status st = pool_checks(grad_out, grad_in, idx, dtype, B, H, W, C);
  if (st != HD_OK) return st;
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long items = (long long)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / (dtype == 0 ? 4 : 8));  // 2 x 2 input blocks
  if (dtype == 0)
    hipLaunchKernelGGL((maxpool3s2_bwd_kernel<float>), dim3(pool_grid(items)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float*>(grad_out), idx, B, H, W, C, Ho, Wo, reinterpret_cast<float*>(grad_in));
  else
    hipLaunchKernelGGL((maxpool3s2_bwd_kernel<__hip_bfloat16>), dim3(pool_grid(items)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const __hip_bfloat16*>(grad_out), idx, B, H, W, C, Ho, Wo,
                       reinterpret_cast<__hip_bfloat16*>(grad_in));
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

