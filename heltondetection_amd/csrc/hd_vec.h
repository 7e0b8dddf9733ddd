// hd_vec.h — 16-byte-per-lane row vectors of a channels-last tensor (4 x f32 or 8 x bf16): the unit every
// memory-bound kernel here loads and stores (one fully coalesced 1 KiB request per wave instruction).
#pragma once
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

template <typename T>
struct RowVec;
template <>
struct RowVec<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void unpack(const uint4 r, float* v) {
    v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y); v[2] = __uint_as_float(r.z); v[3] = __uint_as_float(r.w);
  }
  static __device__ __forceinline__ void load(const float* p, float* v) {
    unpack(*reinterpret_cast<const uint4*>(p), v);
  }
  static __device__ __forceinline__ void store(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <>
struct RowVec<__hip_bfloat16> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void unpack(const uint4 r, float* v) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void load(const __hip_bfloat16* p, float* v) {
    unpack(*reinterpret_cast<const uint4*>(p), v);
  }
  static __device__ __forceinline__ void store(__hip_bfloat16* p, const float* v) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // one v_cvt_pk_bf16_f32 (round-to-nearest-even) per pair
      const f32x2 f = {v[2 * i], v[2 * i + 1]};
      const bf16x2 r = __builtin_convertvector(f, bf16x2);
      w[i] = *reinterpret_cast<const uint32_t*>(&r);
    }
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

