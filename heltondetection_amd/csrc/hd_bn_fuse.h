// hd_bn_fuse.h — BatchNorm + activation BACKWARD folded into the operand load of a convolution gradient kernel.
//
// A layer is  y = conv(x, w);  out = act(bn(y)) [+ r].  Given d(out), the fused-BN backward of csrc/bn_act.hip runs
//   bn_bwd_reduce_kernel   sum dz, sum dz*xhat per channel          (dz = d(out) * act'(z), z = bn(y))
//   bn_bwd_dx_kernel       dy = A*dz + (P*y + Q)                    -> a bf16 tensor the convolution gradients then read
// The second pass is elementwise in (d(out), y) with per-channel coefficients, so the data-gradient and weight-gradient
// kernels can form dy themselves while they stage their dY operand: they load the d(out) chunk and the y chunk (16 bytes
// each), apply exactly the arithmetic of bn_bwd_dx_kernel (same operation order, one round-to-nearest-even to bf16), and
// hand the result to the MFMA path.  That removes one launch and one write + one read of the tensor per layer; the bits
// of dy — and so of dx and dw — are the ones the unfused path produces.
#pragma once
#include "hd_common.h"

namespace hdfuse {

enum { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2 };
constexpr int kRep = 16;  // == hd_bn_act_replicas()

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// the same expressions as csrc/bn_act.hip (fast_sigmoid / act_grad): the two must stay in lockstep
__device__ __forceinline__ float fast_sigmoid(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }
__device__ __forceinline__ float act_grad(int act, float z) {
  if (act == ACT_SILU) {
    const float s = fast_sigmoid(z);
    return fmaf(z * s, 1.0f - s, s);
  }
  if (act == ACT_RELU) return z > 0.0f ? 1.0f : 0.0f;
  return 1.0f;
}

// Device-side view of hd_bn_bwd (+ the second place of d(out), hd_rows2) — passed to the kernels by value.
struct BnBwd {
  const unsigned short* y;  // the BatchNorm's input = the convolution's raw output, rows [R, C] bf16
  long long ld_y;
  const unsigned short* dout2;  // channels >= split of d(out) live here (row stride ld2); nullptr: one place
  long long ld2;
  int split;
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;
  const float* dsums;  // [kRep, 2, C]: sum dz, sum dz*xhat (hd_bn_act_bwd_reduce)
  float inv_rows;
  int act;
  int C;
  float* dgamma;  // written (with dbeta) by the weight-gradient kernel when non-null
  float* dbeta;
};

__device__ __forceinline__ float rep_sum(const float* __restrict__ p, int idx, int C) {
  float v = 0.0f;
#pragma unroll
  for (int r = 0; r < kRep; ++r) v += p[(size_t)r * 2 * C + idx];
  return v;
}

// Coefficients of channel c, as bn_bwd_dx_kernel folds them:  z = az*y + bz,  dy = az*dz + (P*y + Q).
// d0 / d1: the channel's sum dz / sum dz*xhat (= dbeta / dgamma).
__device__ __forceinline__ void coef(const BnBwd& f, int c, float& az, float& bz, float& P, float& Q, float& d0, float& d1) {
  d0 = rep_sum(f.dsums, c, f.C);
  d1 = rep_sum(f.dsums, f.C + c, f.C);
  const float k1 = d0 * f.inv_rows, k2 = d1 * f.inv_rows;
  const float is = f.invstd[c], c0 = -f.mean[c] * is, ga = f.gamma[c];
  az = ga * is;
  bz = fmaf(ga, c0, f.beta[c]);
  P = -az * k2 * is;
  Q = -az * (k1 + k2 * c0);
}

// Fills s_coef[4][n] (az | bz | P | Q) for channels [c0, c0 + n) with the calling workgroup's threads; the caller
// synchronises before reading.  write_params: this workgroup also publishes dgamma / dbeta of those channels.
__device__ __forceinline__ void fill_coefs(const BnBwd& f, int c0, int n, float* s_coef, bool write_params) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float az, bz, P, Q, d0, d1;
    coef(f, c0 + i, az, bz, P, Q, d0, d1);
    s_coef[i] = az;
    s_coef[n + i] = bz;
    s_coef[2 * n + i] = P;
    s_coef[3 * n + i] = Q;
    if (write_params && f.dgamma) {
      f.dbeta[c0 + i] = d0;
      f.dgamma[c0 + i] = d1;
    }
  }
}

struct Coef8 {  // the 8 channels of one 16-byte chunk
  float az[8], bz[8], P[8], Q[8];
  __device__ __forceinline__ void load(const float* s_coef, int n, int ch0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      az[k] = s_coef[ch0 + k];
      bz[k] = s_coef[n + ch0 + k];
      P[k] = s_coef[2 * n + ch0 + k];
      Q[k] = s_coef[3 * n + ch0 + k];
    }
  }
};

// one 16-byte chunk: 8 x (d(out), y) -> 8 x dy, bf16 in, float32 arithmetic, bf16 out (v_cvt_pk_bf16_f32, RNE)
__device__ __forceinline__ u32x4 chunk(const u32x4 g, const u32x4 yv, const Coef8& c, int act) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  u32x4 out;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float y0 = __uint_as_float(yv[i] << 16), y1 = __uint_as_float(yv[i] & 0xffff0000u);
    const float g0 = __uint_as_float(g[i] << 16), g1 = __uint_as_float(g[i] & 0xffff0000u);
    const float z0 = fmaf(c.az[2 * i], y0, c.bz[2 * i]), z1 = fmaf(c.az[2 * i + 1], y1, c.bz[2 * i + 1]);
    const float dz0 = g0 * act_grad(act, z0), dz1 = g1 * act_grad(act, z1);
    const f32x2 r = {fmaf(c.az[2 * i], dz0, fmaf(c.P[2 * i], y0, c.Q[2 * i])),
                     fmaf(c.az[2 * i + 1], dz1, fmaf(c.P[2 * i + 1], y1, c.Q[2 * i + 1]))};
    const bf16x2 b = __builtin_convertvector(r, bf16x2);
    out[i] = __builtin_bit_cast(unsigned, b);
  }
  return out;
}

// d(out) chunk address of (row m, channel ch0) for an operand that may live in two places
struct Place {
  const unsigned short* base;
  long long ld;
  __device__ __forceinline__ void set(const unsigned short* p, long long ld1, const BnBwd& f, int ch0) {
    if (f.dout2 && ch0 >= f.split) {
      base = f.dout2 + (ch0 - f.split);
      ld = f.ld2;
    } else {
      base = p + ch0;
      ld = ld1;
    }
  }
};

// Row-sliced partial products of a weight gradient, waiting for their sum over the slices (hd_wgrad_job).
struct ReduceJob {
  const float* slab;   // [slices][count] float32
  int slices;
  long long count;     // elements of dw (a multiple of 4)
  unsigned short* dw;  // bf16 destination
};

// dW[q] (4 consecutive elements) = sum over the slices; 16 quads x 16 slab lanes per 256-thread block.  `part`: [16][17].
__device__ __forceinline__ void reduce_block(const ReduceJob& j, long long block, float4 (*part)[17]) {
  const int ql = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long long quad = block * 16 + ql;
  const int S = j.slices;
  const long long NK = j.count;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (quad * 4 < NK) {
    int s = sl;
    for (; s + 48 < S; s += 64) {  // four independent loads in flight
      const float4 v0 = *reinterpret_cast<const float4*>(j.slab + (size_t)s * NK + quad * 4);
      const float4 v1 = *reinterpret_cast<const float4*>(j.slab + (size_t)(s + 16) * NK + quad * 4);
      const float4 v2 = *reinterpret_cast<const float4*>(j.slab + (size_t)(s + 32) * NK + quad * 4);
      const float4 v3 = *reinterpret_cast<const float4*>(j.slab + (size_t)(s + 48) * NK + quad * 4);
      a.x += (v0.x + v1.x) + (v2.x + v3.x);
      a.y += (v0.y + v1.y) + (v2.y + v3.y);
      a.z += (v0.z + v1.z) + (v2.z + v3.z);
      a.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; s < S; s += 16) {
      const float4 v = *reinterpret_cast<const float4*>(j.slab + (size_t)s * NK + quad * 4);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  part[sl][ql] = a;
  __syncthreads();
  if (sl == 0 && quad * 4 < NK) {
#pragma unroll
    for (int i = 1; i < 16; ++i) {
      const float4 v = part[i][ql];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    const f32x4 f = {a.x, a.y, a.z, a.w};
    *reinterpret_cast<bf16x4*>(j.dw + quad * 4) = __builtin_convertvector(f, bf16x4);
  }
}

__host__ __device__ inline long long reduce_blocks(long long count) { return (count / 4 + 15) / 16; }

// RB consecutive reduce blocks at once (same arithmetic per block as reduce_block, bit for bit): RB independent loads per
// thread in flight.  For the riders of LOW-occupancy kernels (one workgroup per CU: csrc/pointwise_bwd.hip) — a 3x3 weight
// gradient's job has 16 slices, i.e. ONE 16-byte load per thread per block, 4 KB in flight per CU: 37 MB took 30 us that way.
// `part`: [RB][16][17] float4 of LDS.
template <int RB>
__device__ __forceinline__ void reduce_blocks_multi(const ReduceJob& j, long long first, float4 (*part)[16][17]) {
  const int ql = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int S = j.slices;
  const long long NK = j.count;
  float4 a[RB];
  long long quad[RB];
  bool in[RB];
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    a[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    quad[b] = (first + b) * 16 + ql;
    in[b] = quad[b] * 4 < NK;
    if (!in[b]) quad[b] = 0;   // a valid address; the value is discarded
  }
  int s = sl;
  for (; s + 48 < S; s += 64) {
    float4 v0[RB], v1[RB], v2[RB], v3[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      v0[b] = *reinterpret_cast<const float4*>(j.slab + (size_t)s * NK + quad[b] * 4);
      v1[b] = *reinterpret_cast<const float4*>(j.slab + (size_t)(s + 16) * NK + quad[b] * 4);
      v2[b] = *reinterpret_cast<const float4*>(j.slab + (size_t)(s + 32) * NK + quad[b] * 4);
      v3[b] = *reinterpret_cast<const float4*>(j.slab + (size_t)(s + 48) * NK + quad[b] * 4);
    }
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      a[b].x += (v0[b].x + v1[b].x) + (v2[b].x + v3[b].x);
      a[b].y += (v0[b].y + v1[b].y) + (v2[b].y + v3[b].y);
      a[b].z += (v0[b].z + v1[b].z) + (v2[b].z + v3[b].z);
      a[b].w += (v0[b].w + v1[b].w) + (v2[b].w + v3[b].w);
    }
  }
  for (; s < S; s += 16) {
    float4 v[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) v[b] = *reinterpret_cast<const float4*>(j.slab + (size_t)s * NK + quad[b] * 4);
#pragma unroll
    for (int b = 0; b < RB; ++b) { a[b].x += v[b].x; a[b].y += v[b].y; a[b].z += v[b].z; a[b].w += v[b].w; }
  }
#pragma unroll
  for (int b = 0; b < RB; ++b) part[b][sl][ql] = a[b];
  __syncthreads();
  if (sl == 0) {
#pragma unroll
    for (int b = 0; b < RB; ++b) {
      if (!in[b]) continue;
      float4 t = a[b];
#pragma unroll
      for (int i = 1; i < 16; ++i) {
        const float4 v = part[b][i][ql];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
      typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
      typedef __attribute__((ext_vector_type(4))) float f32x4;
      const f32x4 f = {t.x, t.y, t.z, t.w};
      *reinterpret_cast<bf16x4*>(j.dw + quad[b] * 4) = __builtin_convertvector(f, bf16x4);
    }
  }
}

}  // namespace hdfuse
