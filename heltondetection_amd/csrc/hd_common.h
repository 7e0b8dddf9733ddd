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

// ---- division-free exact quotient ------------------------------------------------------------------
// The IoU kernels are VALU-bound on the IEEE division (v_div_scale x2, v_rcp, 6 FMAs, v_div_fmas, v_div_fixup
// ~ 13 issue slots per output).  hd_quot_fast* produce THE SAME BITS as n / u with v_rcp_f32 + 6 FMA-class
// operations (3 packed slots per output in the two-wide form), for 0 <= n <= u, u > 0, with n == 0 or
// n, u in [2^-86, 2^84] — which the hd_box_div_* predicates guarantee for every pair of boxes they accept
// (tiles holding any other box run hd_iou_full) — EXCEPT when u's significand is all ones: the caller must
// test hd_ones_acc() and recompute such outputs with the IEEE division.  Why the bits are identical
// (exhaustive reciprocal check, Markstein's theorem, exhaustive numerators for the exception denominators)
// is spelled out and machine-checked in tests/proofs/iou_div_proof.c, run by tests/test_iou_div_proof.py.
// The sequence here must stay in lockstep with that file.
typedef float hd_f2 __attribute__((ext_vector_type(2)));

// Coordinates that are +0 or have 2^-20 <= |c| <= 2^40 (NaN, inf and -0 fail: (-0) - (+0) = -0 would give the
// quotient a sign the oracle's `iw > 0 ? iw : 0` never produces).  Any positive overlap width of two such boxes
// is >= 2^-43, so inter >= 2^-86, and union <= 2^84: every intermediate of the sequence is a normal number.
__device__ __forceinline__ bool hd_coord_div_safe(float c) {
  const float a = fabsf(c);
  return __float_as_uint(c) == 0u || (a >= 0x1p-20f && a <= 0x1p40f);
}
// "nn": safe coordinates and w, h >= 0 (area >= 0; all-zero padding boxes qualify).
__device__ __forceinline__ bool hd_box_div_safe_nn(const float4 b) {
  return hd_coord_div_safe(b.x) && hd_coord_div_safe(b.y) && hd_coord_div_safe(b.z) && hd_coord_div_safe(b.w) &&
         b.z >= b.x && b.w >= b.y;
}
// "pos": additionally w, h > 0.  One "pos" box in a pair of "nn" boxes makes the union strictly positive
// (union >= max(area_a, area_b) - rounding of a sum that dominates inter <= min(area_a, area_b)), so the
// fast path needs no `union > 0 ?` select.
__device__ __forceinline__ bool hd_box_div_safe_pos(const float4 b) {
  return hd_coord_div_safe(b.x) && hd_coord_div_safe(b.y) && hd_coord_div_safe(b.z) && hd_coord_div_safe(b.w) &&
         b.z > b.x && b.w > b.y;
}

// v_min_f32 / v_max_f32 without the sNaN-quieting canonicalisation (a v_max x, x, x per operand) that fminf /
// fmaxf cost under the IEEE mode bit: the fast path only ever sees NaN-free boxes.
__device__ __forceinline__ float hd_min_raw(float a, float b) {
  float d;
  asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float hd_max_raw(float a, float b) {
  float d;
  asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// Running detector of "some denominator had an all-ones significand": max over (bits | ~significand mask)
// reaches 0xFFFFFFFF exactly then.  hd_ones_acc(acc) is the test.
__device__ __forceinline__ uint32_t hd_ones_key(float u) { return __float_as_uint(u) | 0xFF800000u; }
__device__ __forceinline__ bool hd_ones_acc(uint32_t acc) { return acc == 0xFFFFFFFFu; }

__device__ __forceinline__ float hd_quot_fast(float n, float u) {
  const float y0 = __builtin_amdgcn_rcpf(u);
  const float e = fmaf(-u, y0, 1.0f);
  const float t = fmaf(e, e, e);
  const float y1 = fmaf(t, y0, y0);
  const float q0 = n * y1;
  const float r = fmaf(-u, q0, n);
  return fmaf(r, y1, q0);
}
__device__ __forceinline__ hd_f2 hd_quot_fast2(hd_f2 n, hd_f2 u) {
  const hd_f2 one = {1.0f, 1.0f};
  const hd_f2 y0 = {__builtin_amdgcn_rcpf(u.x), __builtin_amdgcn_rcpf(u.y)};
  const hd_f2 e = __builtin_elementwise_fma(-u, y0, one);    // v_pk_fma_f32 from here on
  const hd_f2 t = __builtin_elementwise_fma(e, e, e);
  const hd_f2 y1 = __builtin_elementwise_fma(t, y0, y0);
  const hd_f2 q0 = n * y1;
  const hd_f2 r = __builtin_elementwise_fma(-u, q0, n);
  return __builtin_elementwise_fma(r, y1, q0);
}

// hd_iou_full() with the fast quotient: identical bits when one box is "pos", the other "nn", and the
// denominator's significand is not all ones (*ones accumulates the detector; the caller falls back then).
__device__ __forceinline__ float hd_iou_fast(const float4 a, const float area_a, const float4 b, const float area_b,
                                             uint32_t* ones) {
  const float iw = hd_max_raw(hd_min_raw(a.z, b.z) - hd_max_raw(a.x, b.x), 0.0f);
  const float ih = hd_max_raw(hd_min_raw(a.w, b.w) - hd_max_raw(a.y, b.y), 0.0f);
  const float inter = iw * ih;
  const float uni = (area_a + area_b) - inter;
  *ones = max(*ones, hd_ones_key(uni));
  return hd_quot_fast(inter, uni);
}
// two columns at once: the subtractions, products and the whole quotient run as packed f32 operations
__device__ __forceinline__ hd_f2 hd_iou_fast2(const float4 a, const float area_a, const float4 b0, const float area_b0,
                                              const float4 b1, const float area_b1, uint32_t* ones) {
  const hd_f2 x_hi = {hd_min_raw(a.z, b0.z), hd_min_raw(a.z, b1.z)}, x_lo = {hd_max_raw(a.x, b0.x), hd_max_raw(a.x, b1.x)};
  const hd_f2 y_hi = {hd_min_raw(a.w, b0.w), hd_min_raw(a.w, b1.w)}, y_lo = {hd_max_raw(a.y, b0.y), hd_max_raw(a.y, b1.y)};
  const hd_f2 dw = x_hi - x_lo, dh = y_hi - y_lo;
  const hd_f2 iw = {hd_max_raw(dw.x, 0.0f), hd_max_raw(dw.y, 0.0f)};
  const hd_f2 ih = {hd_max_raw(dh.x, 0.0f), hd_max_raw(dh.y, 0.0f)};
  const hd_f2 inter = iw * ih;
  const hd_f2 areas = {area_a + area_b0, area_a + area_b1};
  const hd_f2 uni = areas - inter;
  *ones = max(*ones, max(hd_ones_key(uni.x), hd_ones_key(uni.y)));
  return hd_quot_fast2(inter, uni);
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

// ---- launches with optional kernel-only timing ----------------------------------------------------------
// hd_launch() is hipLaunchKernel with the arguments converted to the kernel's exact parameter types.  When the
// caller armed this thread with hd_profile_time_next_launch(start, stop) the launch goes through hipExtLaunchKernel
// instead, which stamps the two events with the kernel's OWN begin / end (the dispatch packet's timestamps — the
// interval rocprofv3 reports), so bench.py's rooflines are HIP-event measurements of the kernel alone, without
// the host launch gap or the ~1.5 us boundary an event pair around the launch would include.
#include <hip/hip_ext.h>

#include <tuple>
#include <utility>

struct HdLaunchTiming {
  hipEvent_t start = nullptr, stop = nullptr;
};
inline HdLaunchTiming& hd_launch_timing() {
  static thread_local HdLaunchTiming t;
  return t;
}

template <typename Tuple, size_t... I>
inline hipError_t hd_launch_impl(const void* fn, dim3 grid, dim3 block, size_t lds, hipStream_t s, Tuple& args,
                                 std::index_sequence<I...>) {
  void* ptrs[] = {static_cast<void*>(&std::get<I>(args))...};
  HdLaunchTiming& t = hd_launch_timing();
  if (t.stop != nullptr) {
    const hipEvent_t a = t.start, b = t.stop;
    t.start = t.stop = nullptr;  // one launch per arming
    return hipExtLaunchKernel(fn, grid, block, ptrs, lds, s, a, b, 0);
  }
  return hipLaunchKernel(fn, grid, block, ptrs, lds, s);
}
template <typename... KArgs, typename... Args>
inline hipError_t hd_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t s, Args&&... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "hd_launch: argument count does not match the kernel");
  std::tuple<KArgs...> held{static_cast<KArgs>(args)...};
  return hd_launch_impl(reinterpret_cast<const void*>(kernel), grid, block, lds, s, held, std::index_sequence_for<KArgs...>{});
}

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
