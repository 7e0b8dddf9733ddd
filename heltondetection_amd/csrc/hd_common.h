// hd_common.h — shared device helpers for the gfx950 detection kernels.
//
// Everything here is compiled with -ffp-contract=off: the box arithmetic must round exactly
// like oracle/hd_oracle.c (same expression order, no FMA contraction) because NMS keep masks
// and argmax indices are compared bit-for-bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hd_ops.h"

#define HD_WAVE 64

#define HD_RETURN_IF_LAUNCH_FAILED()                 \
  do {                                               \
    if (hipGetLastError() != hipSuccess) return HD_ERR_LAUNCH; \
  } while (0)

static inline bool hd_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline size_t hd_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int hd_next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// IoU of xyxy boxes — expression order is the oracle's iou_xyxy().
__device__ __forceinline__ float hd_box_area(const float4 a) { return (a.z - a.x) * (a.w - a.y); }

__device__ __forceinline__ float hd_iou_full(const float4 a, const float area_a, const float4 b,
                                             const float area_b) {
  float iw = fminf(a.z, b.z) - fmaxf(a.x, b.x);
  float ih = fminf(a.w, b.w) - fmaxf(a.y, b.y);
  iw = iw > 0.0f ? iw : 0.0f;
  ih = ih > 0.0f ? ih : 0.0f;
  float inter = iw * ih;
  float uni = (area_a + area_b) - inter;
  return uni > 0.0f ? inter / uni : 0.0f;
}

// True iff the boxes overlap with positive width and height (then and only then IoU can be > 0).
__device__ __forceinline__ bool hd_overlaps(const float4 a, const float4 b) {
  float iw = fminf(a.z, b.z) - fmaxf(a.x, b.x);
  float ih = fminf(a.w, b.w) - fmaxf(a.y, b.y);
  return (iw > 0.0f) && (ih > 0.0f);
}

__device__ __forceinline__ float hd_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float hd_bce_logits(float x, float z) {
  return fmaxf(x, 0.0f) - x * z + log1pf(expf(-fabsf(x)));
}

// Focal-modulated BCE-with-logits and its derivative w.r.t. the logit (z may be a soft target):
//   L = af * m * bce,  m = q^gamma, q = 1 - p_t, p_t = z p + (1-z)(1-p), af = z a + (1-z)(1-a) (1 if a <= 0)
//   dL/dx = af * (m (p - z) + bce * dm/dx),  dm/dx = -gamma q^(gamma-1) (2z - 1) p (1 - p)
// q^(gamma-1) is formed as m / q and dropped at q == 0, where bce * q^(gamma-1) -> 0 for every gamma > 0.
__device__ __forceinline__ float hd_focal_bce(float x, float z, float gamma, float alpha, float* dldx) {
  const float p = hd_sigmoid(x);
  const float bce = hd_bce_logits(x, z);
  const float q = 1.0f - (z * p + (1.0f - z) * (1.0f - p));
  const float af = alpha > 0.0f ? z * alpha + (1.0f - z) * (1.0f - alpha) : 1.0f;
  const float m = powf(q, gamma);
  const float dm = q > 0.0f ? -gamma * (m / q) * ((2.0f * z - 1.0f) * (p * (1.0f - p))) : 0.0f;
  *dldx = af * (m * (p - z) + bce * dm);
  return af * (m * bce);
}

// order-preserving float -> uint32 (larger float -> larger key)
__device__ __forceinline__ uint32_t hd_f2key(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float hd_key2f(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// 64-lane sum / max reductions (result valid in every lane for xor butterflies)
__device__ __forceinline__ float hd_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int hd_wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ int hd_lane() { return threadIdx.x & 63; }

// Device-side fill (32-bit pattern).  Used instead of hipMemsetAsync: a memset becomes a memset
// NODE when the caller captures the stream into a hipGraph, and on ROCm 7.2 those nodes were
// observed not to stay ordered with the surrounding kernel nodes once eager launches are
// interleaved with replays; a plain kernel is ordered like every other launch and
// costs about the same (~2 us).
static __global__ __launch_bounds__(256) void hd_fill32_kernel(uint32_t* __restrict__ p, uint32_t v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
static inline hd_status hd_fill32(void* p, uint32_t v, size_t n_words, hipStream_t s) {
  if (n_words == 0) return HD_OK;
  size_t blocks = (n_words + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(hd_fill32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<uint32_t*>(p), v, n_words);
  return hipGetLastError() == hipSuccess ? HD_OK : HD_ERR_LAUNCH;
}
