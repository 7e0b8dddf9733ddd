// bn_act.hip — fused BatchNorm2d (training statistics) + activation, forward and backward, for
// dense channels-last tensors viewed as rows [R = B*H*W, C].
//
// Why hand-written (MI355X guide: "fuse elementwise / normalisation / activation work"): on the
// YOLOv5s train step PyTorch-ROCm runs 3 MIOpen BN kernels + 1 SiLU kernel forward and 1 + 3
// backward per layer, 57 layers — 456 launches, ~13 passes over each activation.  Here:
//   forward : bn_stats_kernel (read x)                -> per-channel sum, sum of squares (fp32 atomics)
//             bn_apply_kernel (read x, write y)       -> y = act(a_c*x + b_c); block 0 also updates the
//                                                        running statistics and saves mean / invstd
//   backward: bn_bwd_reduce_kernel (read x, dy)       -> sum dz, sum dz*xhat   (dz = dy * act'(z), z recomputed)
//             bn_bwd_dx_kernel (read x, dy, write dx) -> dx; block 0 writes dgamma / dbeta
// = 4 launches and 8 passes, and the BN output never has to be kept for the activation's backward.
//
// Mapping.  A lane owns V consecutive channels (16 bytes: 8 bf16 or 4 f32) of one row; consecutive
// lanes walk the flat [R*C/V] vector space, so every access is a fully coalesced 16-byte vector and
// a lane's channel group g = i % (C/V) is fixed (the launch keeps blockDim*gridDim a multiple of
// C/V), which lets per-channel coefficients and partial sums live in registers.
// Per-channel sums are combined in LDS (ds_add_f32) and published with one global float atomic per
// channel per workgroup (<= 512 workgroups, so same-address serialisation stays < 3 us).
#include <hip/hip_bf16.h>

#include <type_traits>

#include "hd_common.h"
#include "hd_vec.h"

namespace {

enum { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2 };

// sigmoid via v_exp_f32 + v_rcp_f32 (1 ulp).  An IEEE division costs ~14 instruction slots here (the
// library is built with -ffp-contract=off and no fast-math for the box kernels' sake), which made the
// SiLU layers VALU-bound: ~100 cycles per element-wave against ~40 with the reciprocal.  The results are
// rounded to bf16 / compared with PyTorch at 1e-2..1e-3, far above 1 ulp of fp32.
__device__ __forceinline__ float fast_sigmoid(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }

template <int ACT>
__device__ __forceinline__ float act_fwd(float z) {
  if (ACT == ACT_SILU) return z * fast_sigmoid(z);
  if (ACT == ACT_RELU) return z > 0.0f ? z : 0.0f;
  return z;
}
// d act(z) / dz
template <int ACT>
__device__ __forceinline__ float act_grad(float z) {
  if (ACT == ACT_SILU) {
    const float s = fast_sigmoid(z);
    return fmaf(z * s, 1.0f - s, s);  // s + z s (1 - s)
  }
  if (ACT == ACT_RELU) return z > 0.0f ? 1.0f : 0.0f;
  return 1.0f;
}

constexpr int kMaxC = 4096;
// Per-channel accumulators are replicated kRep times ([kRep, 2, C]); a workgroup adds into replica
// blockIdx % kRep (= its XCD under round-robin dispatch).  With one copy, up to 512 workgroups end
// with atomics onto the same 2*C addresses, which the memory-side atomic units serialise (~14x
// below the spread-out rate, MI355X_MICROARCH 'Global float atomics'): a 22-25 us floor per launch.
constexpr int kRep = 16;

// Workgroup fold of per-lane channel partials a[V], b[V] (lane's channel group g, cg groups) into
// global rows dst[0..C) and dst[C..2C).  Lanes of a wave that own the same group (cg < 64: lane % cg)
// are first combined with a shuffle butterfly, each wave then parks its cg x V results in its own LDS
// slot with plain stores, and the slots are summed while publishing: no LDS atomics.  (With
// ds_add_f32 straight from every lane, C = 32 means 4 groups -> 16 lanes of a wave on one LDS address
// for each of 2V instructions, a 16-way serialisation per workgroup that set the reduce kernels' floor.)
// Requires blockDim.x == 256 and cg a power of two <= 256; other shapes use fold_atomic().
template <int V>
__device__ __forceinline__ void fold_slots(float* s_part /* [4][2C] */, float* a, float* b, int g, int cg, int C,
                                           float* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  for (int off = cg; off < 64; off <<= 1) {
#pragma unroll
    for (int k = 0; k < V; ++k) {
      a[k] += __shfl_xor(a[k], off, 64);
      b[k] += __shfl_xor(b[k], off, 64);
    }
  }
  const int span = cg < 64 ? 64 : cg;        // threads per full sweep of the groups
  const int nslots = 256 / span;             // 4, 2 or 1
  const int slot = threadIdx.x / span;
  if (cg >= 64 || lane < cg) {
    float* row = s_part + (size_t)slot * 2 * C;
#pragma unroll
    for (int k = 0; k < V; ++k) {
      row[g * V + k] = a[k];
      row[C + g * V + k] = b[k];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    float v = s_part[i];
    for (int sl = 1; sl < nslots; ++sl) v += s_part[(size_t)sl * 2 * C + i];
    atomicAdd(&dst[i], v);
  }
}

template <int V>
__device__ __forceinline__ void fold_atomic(float* s_acc /* [2C], zeroed */, const float* a, const float* b, int g,
                                            int C, float* __restrict__ dst) {
#pragma unroll
  for (int k = 0; k < V; ++k) {
    atomicAdd(&s_acc[g * V + k], a[k]);
    atomicAdd(&s_acc[C + g * V + k], b[k]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) atomicAdd(&dst[i], s_acc[i]);
}

__device__ __forceinline__ bool fold_by_slots(int cg) { return blockDim.x == 256 && (cg & (cg - 1)) == 0; }

// ---- forward statistics -------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, long long nvec, int C,
                                                       float* __restrict__ sums /* [kRep, 2, C] */) {
  constexpr int V = RowVec<T>::N;
  extern __shared__ float s_acc[];  // [4][2*C] wave slots (fold_slots) or [2*C] accumulators (fold_atomic)
  const int cg = C / V;
  const bool slots = fold_by_slots(cg);
  if (!slots) {
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) s_acc[i] = 0.0f;
    __syncthreads();
  }
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = (int)(i0 % cg);
  float s[V], ss[V];
#pragma unroll
  for (int k = 0; k < V; ++k) s[k] = ss[k] = 0.0f;
  long long i = i0;
  for (; i + 3 * stride < nvec; i += 4 * stride) {  // 4 independent 16-byte loads in flight per lane
    uint4 raw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const uint4*>(x + (i + u * stride) * V);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float v[V];
      RowVec<T>::unpack(raw[u], v);
#pragma unroll
      for (int k = 0; k < V; ++k) {
        s[k] += v[k];
        ss[k] = fmaf(v[k], v[k], ss[k]);
      }
    }
  }
  for (; i < nvec; i += stride) {
    float v[V];
    RowVec<T>::load(x + i * V, v);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      s[k] += v[k];
      ss[k] = fmaf(v[k], v[k], ss[k]);
    }
  }
  float* dst = sums + (size_t)(blockIdx.x % kRep) * 2 * C;
  if (slots) fold_slots<V>(s_acc, s, ss, g, cg, C, dst);
  else fold_atomic<V>(s_acc, s, ss, g, C, dst);
}

__device__ __forceinline__ float rep_sum(const float* __restrict__ p, int idx, int C) {
  float v = 0.0f;
#pragma unroll
  for (int r = 0; r < kRep; ++r) v += p[(size_t)r * 2 * C + idx];
  return v;
}

// A [R, C] channels-last operand that may live in TWO places: channel groups [0, split) in rows of stride ld at p,
// groups [split, cg) in rows of stride ld2 at p2.  One place, dense (ld == C, split == cg) is the fast case with no
// division.  A strided single place is a channel slice of a wider channels-last tensor (the gradient of a torch.cat
// input / the slot of a concatenation buffer); two places serve the merged C3 branch convolutions, whose first half
// feeds the bottleneck stack as a dense tensor while the second half goes straight into the concatenation.
template <typename T>
struct Rows {
  T* p;
  long long ld;
  T* p2;
  long long ld2;
  int split;  // in vector groups
  __device__ __forceinline__ T* at(long long i, int cg, int C) const {
    constexpr int V = RowVec<typename std::remove_const<T>::type>::N;
    if (ld == C && split == cg) return p + i * V;
    const uint32_t r = (uint32_t)i / (uint32_t)cg;
    const uint32_t g = (uint32_t)i - r * (uint32_t)cg;
    return (int)g < split ? p + (long long)r * ld + (long long)g * V : p2 + (long long)r * ld2 + (long long)((int)g - split) * V;
  }
};

// ---- forward apply ------------------------------------------------------------------------------
// RES: 0 no residual; 1 y = act(bn(x) + r) (ResNet bottleneck tail); 2 y = act(bn(x)) + r (YOLO bottleneck)
template <typename T, int ACT, int RES>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ res, long long nvec,
                                                       int C, float inv_R,
                                                       float unbias, const float* __restrict__ sums,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, float momentum,
                                                       float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, float* __restrict__ save_mean,
                                                       float* __restrict__ save_invstd, const Rows<T> y) {
  constexpr int V = RowVec<T>::N;
  extern __shared__ float s_coef[];  // [2*C]: a_c, b_c — computed once per workgroup from the replicas
  const int cg = C / V;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = (int)(i0 % cg);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float mean = rep_sum(sums, c, C) * inv_R;
    const float var = fmaxf(rep_sum(sums, C + c, C) * inv_R - mean * mean, 0.0f);
    const float invstd = rsqrtf(var + eps);
    const float ac = gamma[c] * invstd;
    s_coef[c] = ac;
    s_coef[C + c] = beta[c] - mean * ac;
    if (blockIdx.x == 0) {
      save_mean[c] = mean;
      save_invstd[c] = invstd;
      if (running_mean) {
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * var * unbias;
      }
    }
  }
  __syncthreads();
  float a[V], b[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    a[k] = s_coef[g * V + k];
    b[k] = s_coef[C + g * V + k];
  }
  long long i = i0;
  for (; i + 3 * stride < nvec; i += 4 * stride) {  // 4 independent 16-byte loads in flight per lane
    float v[4][V];
#pragma unroll
    for (int u = 0; u < 4; ++u) RowVec<T>::load(x + (i + u * stride) * V, v[u]);
    if (RES == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int k = 0; k < V; ++k) v[u][k] = act_fwd<ACT>(fmaf(a[k], v[u][k], b[k]));
        RowVec<T>::store(y.at(i + u * stride, cg, C), v[u]);
      }
    } else {
      float rv[4][V];
#pragma unroll
      for (int u = 0; u < 4; ++u) RowVec<T>::load(res + (i + u * stride) * V, rv[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
          const float z = fmaf(a[k], v[u][k], b[k]);
          v[u][k] = RES == 1 ? act_fwd<ACT>(z + rv[u][k]) : act_fwd<ACT>(z) + rv[u][k];
        }
        RowVec<T>::store(y.at(i + u * stride, cg, C), v[u]);
      }
    }
  }
  for (; i < nvec; i += stride) {
    float v[V], rv[V];
    RowVec<T>::load(x + i * V, v);
    if (RES != 0) RowVec<T>::load(res + i * V, rv);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float z = fmaf(a[k], v[k], b[k]);
      v[k] = RES == 0 ? act_fwd<ACT>(z) : (RES == 1 ? act_fwd<ACT>(z + rv[k]) : act_fwd<ACT>(z) + rv[k]);
    }
    RowVec<T>::store(y.at(i, cg, C), v);
  }
}

// inference: running statistics, no reduction
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_infer_kernel(const T* __restrict__ x, long long nvec, int C,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       const float* __restrict__ running_mean,
                                                       const float* __restrict__ running_var, float eps,
                                                       const Rows<T> y) {
  constexpr int V = RowVec<T>::N;
  const int cg = C / V;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = (int)(i0 % cg);
  float a[V], b[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const int c = g * V + k;
    a[k] = gamma[c] * rsqrtf(running_var[c] + eps);
    b[k] = beta[c] - running_mean[c] * a[k];
  }
  for (long long i = i0; i < nvec; i += stride) {
    float v[V];
    RowVec<T>::load(x + i * V, v);
#pragma unroll
    for (int k = 0; k < V; ++k) v[k] = act_fwd<ACT>(fmaf(a[k], v[k], b[k]));
    RowVec<T>::store(y.at(i, cg, C), v);
  }
}

// ---- backward reductions ------------------------------------------------------------------------
// RES (backward kernels): 0, or 1 = the forward added `res` before the activation, so act' is taken at bn(x) + res
template <typename T, int ACT, int RES>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ x, const Rows<const T> dy,
                                                            const T* __restrict__ res, long long nvec,
                                                            int C,
                                                            const float* __restrict__ save_mean,
                                                            const float* __restrict__ save_invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            float* __restrict__ dsums /* [kRep,2,C]: dbeta, dgamma */) {
  constexpr int V = RowVec<T>::N;
  extern __shared__ float s_acc[];
  const int cg = C / V;
  const bool slots = fold_by_slots(cg);
  if (!slots) {
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) s_acc[i] = 0.0f;
    __syncthreads();
  }
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = (int)(i0 % cg);
  // per-channel constants folded so an element costs two FMAs besides act': xhat = is*x + c0, z = ga*xhat + be
  float is[V], c0[V], ga[V], be[V], s1[V], s2[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const int c = g * V + k;
    is[k] = save_invstd[c];
    c0[k] = -save_mean[c] * is[k];
    ga[k] = gamma[c];
    be[k] = beta[c];
    s1[k] = s2[k] = 0.0f;
  }
  long long i = i0;
  for (; i + stride < nvec; i += 2 * stride) {  // 4 independent 16-byte loads in flight per lane
    uint4 rx[2], rg[2], rr[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      rx[u] = *reinterpret_cast<const uint4*>(x + (i + u * stride) * V);
      rg[u] = *reinterpret_cast<const uint4*>(dy.at(i + u * stride, cg, C));
      if (RES) rr[u] = *reinterpret_cast<const uint4*>(res + (i + u * stride) * V);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float xv[V], gv[V], rv[V];
      RowVec<T>::unpack(rx[u], xv);
      RowVec<T>::unpack(rg[u], gv);
      if (RES) RowVec<T>::unpack(rr[u], rv);
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const float xh = fmaf(is[k], xv[k], c0[k]);
        float z = fmaf(ga[k], xh, be[k]);
        if (RES) z += rv[k];
        const float dz = gv[k] * act_grad<ACT>(z);
        s1[k] += dz;
        s2[k] = fmaf(dz, xh, s2[k]);
      }
    }
  }
  for (; i < nvec; i += stride) {
    float xv[V], gv[V], rv[V];
    RowVec<T>::load(x + i * V, xv);
    RowVec<T>::load(dy.at(i, cg, C), gv);
    if (RES) RowVec<T>::load(res + i * V, rv);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float xh = fmaf(is[k], xv[k], c0[k]);
      float z = fmaf(ga[k], xh, be[k]);
      if (RES) z += rv[k];
      const float dz = gv[k] * act_grad<ACT>(z);
      s1[k] += dz;
      s2[k] = fmaf(dz, xh, s2[k]);
    }
  }
  float* dst = dsums + (size_t)(blockIdx.x % kRep) * 2 * C;
  if (slots) fold_slots<V>(s_acc, s1, s2, g, cg, C, dst);
  else fold_atomic<V>(s_acc, s1, s2, g, C, dst);
}

template <typename T, int ACT, int RES>
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const T* __restrict__ x, const Rows<const T> dy,
                                                        const T* __restrict__ res, T* __restrict__ dres,
                                                        long long nvec, int C, float inv_R,
                                                        const float* __restrict__ save_mean,
                                                        const float* __restrict__ save_invstd,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        const float* __restrict__ dsums, T* __restrict__ dx,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta) {
  constexpr int V = RowVec<T>::N;
  const int cg = C / V;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = (int)(i0 % cg);
  extern __shared__ float s_coef[];  // [2*C]: mean of dz, mean of dz*xhat (replicas summed once per workgroup)
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float d0 = rep_sum(dsums, c, C), d1 = rep_sum(dsums, C + c, C);
    s_coef[c] = d0 * inv_R;
    s_coef[C + c] = d1 * inv_R;
    if (blockIdx.x == 0) {
      dbeta[c] = d0;
      dgamma[c] = d1;
    }
  }
  __syncthreads();
  // dx = ga*is*(dz - k1 - xhat*k2) with xhat = is*x + c0 and z = ga*xhat + be, folded per channel into
  //   z = az*x + bz,   dx = A*dz + (P*x + Q):   three FMAs per element besides act'
  float az[V], bz[V], A[V], P[V], Q[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const int c = g * V + k;
    const float is = save_invstd[c], c0 = -save_mean[c] * is, ga = gamma[c];
    az[k] = ga * is;
    bz[k] = fmaf(ga, c0, beta[c]);
    A[k] = ga * is;
    P[k] = -A[k] * s_coef[C + c] * is;
    Q[k] = -A[k] * (s_coef[c] + s_coef[C + c] * c0);
  }
  long long i = i0;
  for (; i + stride < nvec; i += 2 * stride) {  // 4 independent 16-byte loads in flight per lane
    float xv[2][V], gv[2][V], rv[2][V];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      RowVec<T>::load(x + (i + u * stride) * V, xv[u]);
      RowVec<T>::load(dy.at(i + u * stride, cg, C), gv[u]);
      if (RES) RowVec<T>::load(res + (i + u * stride) * V, rv[u]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int k = 0; k < V; ++k) {
        float z = fmaf(az[k], xv[u][k], bz[k]);
        if (RES) z += rv[u][k];
        const float dz = gv[u][k] * act_grad<ACT>(z);
        xv[u][k] = fmaf(A[k], dz, fmaf(P[k], xv[u][k], Q[k]));
        if (RES) rv[u][k] = dz;  // d(res) = dz
      }
      RowVec<T>::store(dx + (i + u * stride) * V, xv[u]);
      if (RES) RowVec<T>::store(dres + (i + u * stride) * V, rv[u]);
    }
  }
  for (; i < nvec; i += stride) {
    float xv[V], gv[V], rv[V];
    RowVec<T>::load(x + i * V, xv);
    RowVec<T>::load(dy.at(i, cg, C), gv);
    if (RES) RowVec<T>::load(res + i * V, rv);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float z = fmaf(az[k], xv[k], bz[k]);
      if (RES) z += rv[k];
      const float dz = gv[k] * act_grad<ACT>(z);
      xv[k] = fmaf(A[k], dz, fmaf(P[k], xv[k], Q[k]));
      if (RES) rv[k] = dz;
    }
    RowVec<T>::store(dx + i * V, xv);
    if (RES) RowVec<T>::store(dres + i * V, rv);
  }
}

// ---- bias + activation of a biased convolution (the RPN head's conv -> ReLU) ---------------------------------------
// y = act(x + bias_c), in place or not: the convolution is called WITHOUT its bias, so the library's broadcast add, the
// separate activation and — backward — the activation's mask pass and the strided bias-gradient reduction (four passes +
// a slow reduce on a 354 MB level of config 3) become one pass each way.
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bias_act_fwd_kernel(const T* __restrict__ x, long long nvec, int C,
                                                           const float* __restrict__ bias, T* __restrict__ y) {
  constexpr int V = RowVec<T>::N;
  const int cg = C / V;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = (int)(i0 % cg);
  float b[V];
#pragma unroll
  for (int k = 0; k < V; ++k) b[k] = bias[g * V + k];
  long long i = i0;
  for (; i + 3 * stride < nvec; i += 4 * stride) {
    uint4 raw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const uint4*>(x + (i + u * stride) * V);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float v[V];
      RowVec<T>::unpack(raw[u], v);
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] = act_fwd<ACT>(v[k] + b[k]);
      RowVec<T>::store(y + (i + u * stride) * V, v);
    }
  }
  for (; i < nvec; i += stride) {
    float v[V];
    RowVec<T>::load(x + i * V, v);
#pragma unroll
    for (int k = 0; k < V; ++k) v[k] = act_fwd<ACT>(v[k] + b[k]);
    RowVec<T>::store(y + i * V, v);
  }
}

// dx = dy * act'(y) from the OUTPUT y (ReLU: y > 0; none: 1 and no dx is written), d(bias) = per-channel sum of dx into
// the replicated accumulators [kRep, 2, C] (first half; the second half receives zeros) like the BatchNorm reductions.
template <typename T, int ACT, bool WRITE>
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const T* __restrict__ y, const T* __restrict__ dy, long long nvec,
                                                           int C, T* __restrict__ dx, float* __restrict__ dsums) {
  constexpr int V = RowVec<T>::N;
  extern __shared__ float s_acc[];
  const int cg = C / V;
  const bool slots = fold_by_slots(cg);
  if (!slots) {
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) s_acc[i] = 0.0f;
    __syncthreads();
  }
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = (int)(i0 % cg);
  float s1[V], s2[V];
#pragma unroll
  for (int k = 0; k < V; ++k) s1[k] = s2[k] = 0.0f;
  long long i = i0;
  for (; i + stride < nvec; i += 2 * stride) {
    uint4 ry[2], rg[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (ACT != ACT_NONE) ry[u] = *reinterpret_cast<const uint4*>(y + (i + u * stride) * V);
      rg[u] = *reinterpret_cast<const uint4*>(dy + (i + u * stride) * V);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float yv[V], gv[V];
      if (ACT != ACT_NONE) RowVec<T>::unpack(ry[u], yv);
      RowVec<T>::unpack(rg[u], gv);
#pragma unroll
      for (int k = 0; k < V; ++k) {
        if (ACT == ACT_RELU) gv[k] = yv[k] > 0.0f ? gv[k] : 0.0f;
        s1[k] += gv[k];
      }
      if (WRITE) RowVec<T>::store(dx + (i + u * stride) * V, gv);
    }
  }
  for (; i < nvec; i += stride) {
    float yv[V], gv[V];
    if (ACT != ACT_NONE) RowVec<T>::load(y + i * V, yv);
    RowVec<T>::load(dy + i * V, gv);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      if (ACT == ACT_RELU) gv[k] = yv[k] > 0.0f ? gv[k] : 0.0f;
      s1[k] += gv[k];
    }
    if (WRITE) RowVec<T>::store(dx + i * V, gv);
  }
  float* dst = dsums + (size_t)(blockIdx.x % kRep) * 2 * C;
  if (slots) fold_slots<V>(s_acc, s1, s2, g, cg, C, dst);
  else fold_atomic<V>(s_acc, s1, s2, g, C, dst);
}

struct Launch {
  unsigned grid, block;
};

// blockDim and gridDim*blockDim must be multiples of C/V so a lane keeps one channel group
inline Launch plan(long long nvec, int cg, int max_blocks) {
  Launch l;
  l.block = 256 - (256 % cg);  // cg <= 256 (checked by the callers)
  long long need = (nvec + l.block - 1) / l.block;
  long long g = need < max_blocks ? need : max_blocks;
  if (g < 1) g = 1;
  l.grid = (unsigned)g;
  return l;
}

// dynamic LDS of the reduce kernels: wave slots [256/max(cg,64)][2C] (fold_slots) or [2C] (fold_atomic)
inline size_t reduce_lds_bytes(const Launch& l, int cg, int C) {
  const bool slots = l.block == 256 && (cg & (cg - 1)) == 0;
  const int nslots = slots ? 256 / (cg < 64 ? 64 : cg) : 1;
  return (size_t)nslots * 2 * (size_t)C * sizeof(float);
}

inline hd_status common_checks(int dtype, long long R, int C) {
  if ((dtype != 0 && dtype != 1) || R <= 0 || C <= 0 || C > kMaxC) return HD_ERR_SHAPE;
  const int V = dtype == 0 ? 4 : 8;
  if (C % V != 0 || C / V > 256) return HD_ERR_SHAPE;  // one workgroup row must hold every channel group
  return HD_OK;
}

#define HD_ACT_DISPATCH(T, FN, ...)                      \
  do {                                                   \
    if (act == ACT_SILU) FN(T, ACT_SILU, __VA_ARGS__);   \
    else if (act == ACT_RELU) FN(T, ACT_RELU, __VA_ARGS__); \
    else FN(T, ACT_NONE, __VA_ARGS__);                   \
  } while (0)

}  // namespace

// Host side of Rows<T>: primary pointer + row stride (0 or C = dense) and the optional second place (hd_rows2).
template <typename XT>
static hd_status make_rows(XT* p, int64_t row_stride, const hd_rows2* tail, int C, int V, long long nvec, Rows<XT>* out) {
  int64_t ld = row_stride == 0 ? C : row_stride;
  int split = C;
  XT* p2 = nullptr;
  int64_t ld2 = 0;
  if (tail && tail->p2) {
    split = tail->split;
    if (split <= 0 || split >= C || split % V != 0) return HD_ERR_SHAPE;
    p2 = reinterpret_cast<XT*>(const_cast<void*>(static_cast<const void*>(tail->p2)));
    ld2 = tail->ld2 == 0 ? (C - split) : tail->ld2;
    if (ld2 < C - split || ld2 % V != 0) return HD_ERR_SHAPE;
    if (!hd_aligned16(p2)) return HD_ERR_ALIGN;
    if (row_stride == 0) ld = split;  // a dense first place holds only its own channels
    if (ld < split) return HD_ERR_SHAPE;
  } else if (ld < C) {
    return HD_ERR_SHAPE;
  }
  if (ld % V != 0) return HD_ERR_SHAPE;
  if ((ld != C || split != C) && nvec >= (1ll << 31)) return HD_ERR_SHAPE;  // the strided path indexes with 32-bit rows
  out->p = p;
  out->ld = ld;
  out->p2 = p2;
  out->ld2 = ld2;
  out->split = split / V;
  return HD_OK;
}

// dy_row_stride: elements between consecutive rows of dy (0 or C = dense; > C = channel slice of a wider tensor)
static hd_status check_dy_stride(int64_t dy_row_stride, int C, int V, long long nvec, int64_t* ld) {
  *ld = dy_row_stride == 0 ? C : dy_row_stride;
  if (*ld < C || *ld % V != 0) return HD_ERR_SHAPE;
  if (*ld != C && nvec >= (1ll << 31)) return HD_ERR_SHAPE;  // the strided path indexes with 32-bit row numbers
  return HD_OK;
}

extern "C" hd_status hd_bn_act_stats(const void* x, int dtype, int64_t R, int C, float* sums, hd_stream stream) {
  hd_status st = common_checks(dtype, R, C);
  if (st != HD_OK) return st;
  if (!x || !sums) return HD_ERR_NULL;
  if (!hd_aligned16(x)) return HD_ERR_ALIGN;
  const int V = dtype == 0 ? 4 : 8;
  const long long nvec = R * (long long)(C / V);
  // ~8 vectors per lane, <= 2 workgroups per CU (re-swept inside the train step, where x arrives cold for the backward
  // pair: 4 / 2 vectors per lane and 1024 / 2048 workgroups are all within 3 % or worse, profiles/r02_bn_in_step.md)
  const Launch l = plan(nvec / 8 + 1, C / V, 512);
  const size_t lds = reduce_lds_bytes(l, C / V, C);
  if (dtype == 0)
    (void)hd_launch(bn_stats_kernel<float>, dim3(l.grid), dim3(l.block), lds, (hipStream_t)stream,
                       reinterpret_cast<const float*>(x), nvec, C, sums);
  else
    (void)hd_launch(bn_stats_kernel<__hip_bfloat16>, dim3(l.grid), dim3(l.block), lds, (hipStream_t)stream,
                       reinterpret_cast<const __hip_bfloat16*>(x), nvec, C, sums);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

// stat_rows: number of rows the sums cover (0 = R).  Larger than R when the caller all-reduced the sums across
// ranks between _stats and _apply (synchronised BatchNorm): mean/var then are the global ones.
extern "C" hd_status hd_bn_act_apply(const void* x, int dtype, int64_t R, int64_t stat_rows, int C, const float* sums,
                                     const float* gamma, const float* beta, float eps, float momentum,
                                     float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                                     int act, const void* res, int res_mode, void* y, int64_t y_row_stride,
                                     const hd_rows2* y_tail, hd_stream stream) {
  hd_status st = common_checks(dtype, R, C);
  if (st != HD_OK) return st;
  if (!x || !sums || !gamma || !beta || !save_mean || !save_invstd || !y) return HD_ERR_NULL;
  if (res_mode < 0 || res_mode > 2) return HD_ERR_SHAPE;
  if ((res_mode != 0) != (res != nullptr)) return HD_ERR_NULL;
  if (res && !hd_aligned16(res)) return HD_ERR_ALIGN;
  if ((running_mean == nullptr) != (running_var == nullptr)) return HD_ERR_NULL;
  if (!hd_aligned16(x) || !hd_aligned16(y)) return HD_ERR_ALIGN;
  const int V = dtype == 0 ? 4 : 8;
  const long long nvec = R * (long long)(C / V);
  const Launch l = plan(nvec / 4 + 1, C / V, 2048);  // >= 4 vectors per lane: the per-workgroup prologue is not free
  const int64_t RS = stat_rows > 0 ? stat_rows : R;
  const float inv_R = 1.0f / (float)RS;
  const float unbias = RS > 1 ? (float)RS / (float)(RS - 1) : 1.0f;
  hipStream_t s = (hipStream_t)stream;
#define HD_APPLY_R(T, A, RS_, XT)                                                                                  \
  do {                                                                                                             \
    Rows<XT> yr;                                                                                                   \
    st = make_rows<XT>(reinterpret_cast<XT*>(y), y_row_stride, y_tail, C, V, nvec, &yr);                          \
    if (st != HD_OK) return st;                                                                                    \
    (void)hd_launch(bn_apply_kernel<XT, A, RS_>, dim3(l.grid), dim3(l.block), 2 * (size_t)C * sizeof(float), s,    \
                    reinterpret_cast<const XT*>(x), reinterpret_cast<const XT*>(res), nvec, C, inv_R, unbias,      \
                    sums, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, yr);      \
  } while (0)
#define HD_APPLY(T, A, XT)                     \
  do {                                         \
    if (res_mode == 0) HD_APPLY_R(T, A, 0, XT); \
    else if (res_mode == 1) HD_APPLY_R(T, A, 1, XT); \
    else HD_APPLY_R(T, A, 2, XT);              \
  } while (0)
  if (dtype == 0) HD_ACT_DISPATCH(float, HD_APPLY, float);
  else HD_ACT_DISPATCH(__hip_bfloat16, HD_APPLY, __hip_bfloat16);
#undef HD_APPLY
#undef HD_APPLY_R
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_bn_act_infer(const void* x, int dtype, int64_t R, int C, const float* gamma,
                                     const float* beta, const float* running_mean, const float* running_var,
                                     float eps, int act, void* y, int64_t y_row_stride, const hd_rows2* y_tail,
                                     hd_stream stream) {
  hd_status st = common_checks(dtype, R, C);
  if (st != HD_OK) return st;
  if (!x || !gamma || !beta || !running_mean || !running_var || !y) return HD_ERR_NULL;
  if (!hd_aligned16(x) || !hd_aligned16(y)) return HD_ERR_ALIGN;
  const int V = dtype == 0 ? 4 : 8;
  const long long nvec = R * (long long)(C / V);
  const Launch l = plan(nvec / 4 + 1, C / V, 2048);  // >= 4 vectors per lane: the per-workgroup prologue is not free
  hipStream_t s = (hipStream_t)stream;
#define HD_INFER(T, A, XT)                                                                                         \
  do {                                                                                                             \
    Rows<XT> yr;                                                                                                   \
    st = make_rows<XT>(reinterpret_cast<XT*>(y), y_row_stride, y_tail, C, V, nvec, &yr);                          \
    if (st != HD_OK) return st;                                                                                    \
    (void)hd_launch(bn_infer_kernel<XT, A>, dim3(l.grid), dim3(l.block), 0, s, reinterpret_cast<const XT*>(x),     \
                    nvec, C, gamma, beta, running_mean, running_var, eps, yr);                                     \
  } while (0)
  if (dtype == 0) HD_ACT_DISPATCH(float, HD_INFER, float);
  else HD_ACT_DISPATCH(__hip_bfloat16, HD_INFER, __hip_bfloat16);
#undef HD_INFER
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_bn_act_bwd_reduce(const void* x, const void* dy, int64_t dy_row_stride, int dtype, int64_t R,
                                          int C, const float* save_mean, const float* save_invstd, const float* gamma,
                                          const float* beta, int act, const void* res, float* dsums,
                                          const hd_rows2* dy_tail, hd_stream stream) {
  hd_status st = common_checks(dtype, R, C);
  if (st != HD_OK) return st;
  if (!x || !dy || !save_mean || !save_invstd || !gamma || !beta || !dsums) return HD_ERR_NULL;
  if (!hd_aligned16(x) || !hd_aligned16(dy) || (res && !hd_aligned16(res))) return HD_ERR_ALIGN;
  const int V = dtype == 0 ? 4 : 8;
  const long long nvec = R * (long long)(C / V);
  // ~4 vectors per lane, <= 2 workgroups per CU.  Re-swept in the YOLOv5s step of round 5's final build (alternating runs on one
  // box, profiles/r05b_bn_plan_scan.log): 8 vectors per lane (the forward statistics' choice, 100 workgroups for a 3.3 MB
  // layer) 3315 images/s, 4 / 3 vectors 3348 / 3347 (+1.0 %), 6 vectors 3323, 2 vectors on <= 2048 workgroups 3299, 16 on
  // <= 256 3187; caps of 256 / 768 workgroups at 4 vectors 3278 / 3337.  Most layers are small enough that the cap does not
  // bind, so the halved per-lane loop is what counts; the layers of config 3 (100 - 350 MB) sit on the cap either way.
  const Launch l = plan(nvec / 4 + 1, C / V, 512);
  const size_t lds = reduce_lds_bytes(l, C / V, C);
  hipStream_t s = (hipStream_t)stream;
#define HD_RED_R(T, A, RS_, XT)                                                                                    \
  do {                                                                                                             \
    Rows<const XT> dr;                                                                                             \
    st = make_rows<const XT>(reinterpret_cast<const XT*>(dy), dy_row_stride, dy_tail, C, V, nvec, &dr);           \
    if (st != HD_OK) return st;                                                                                    \
    (void)hd_launch(bn_bwd_reduce_kernel<XT, A, RS_>, dim3(l.grid), dim3(l.block), lds, s,                         \
                    reinterpret_cast<const XT*>(x), dr, reinterpret_cast<const XT*>(res), nvec, C, save_mean,      \
                    save_invstd, gamma, beta, dsums);                                                              \
  } while (0)
#define HD_RED(T, A, XT)                \
  do {                                  \
    if (res) HD_RED_R(T, A, 1, XT);     \
    else HD_RED_R(T, A, 0, XT);         \
  } while (0)
  if (dtype == 0) HD_ACT_DISPATCH(float, HD_RED, float);
  else HD_ACT_DISPATCH(__hip_bfloat16, HD_RED, __hip_bfloat16);
#undef HD_RED
#undef HD_RED_R
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_bn_act_bwd_dx(const void* x, const void* dy, int64_t dy_row_stride, int dtype, int64_t R,
                                      int64_t stat_rows, int C, const float* save_mean, const float* save_invstd, const float* gamma,
                                      const float* beta, int act, const void* res, const float* dsums, void* dx,
                                      void* dres, float* dgamma, float* dbeta, const hd_rows2* dy_tail, hd_stream stream) {
  hd_status st = common_checks(dtype, R, C);
  if (st != HD_OK) return st;
  if (!x || !dy || !save_mean || !save_invstd || !gamma || !beta || !dsums || !dx || !dgamma || !dbeta)
    return HD_ERR_NULL;
  if ((res == nullptr) != (dres == nullptr)) return HD_ERR_NULL;
  if (!hd_aligned16(x) || !hd_aligned16(dy) || !hd_aligned16(dx) || (res && (!hd_aligned16(res) || !hd_aligned16(dres))))
    return HD_ERR_ALIGN;
  const int V = dtype == 0 ? 4 : 8;
  const long long nvec = R * (long long)(C / V);
  const Launch l = plan(nvec / 4 + 1, C / V, 2048);  // >= 4 vectors per lane: the per-workgroup prologue is not free
  const float inv_R = 1.0f / (float)(stat_rows > 0 ? stat_rows : R);
  hipStream_t s = (hipStream_t)stream;
#define HD_DX_R(T, A, RS_, XT)                                                                                     \
  do {                                                                                                             \
    Rows<const XT> dr;                                                                                             \
    st = make_rows<const XT>(reinterpret_cast<const XT*>(dy), dy_row_stride, dy_tail, C, V, nvec, &dr);           \
    if (st != HD_OK) return st;                                                                                    \
    (void)hd_launch(bn_bwd_dx_kernel<XT, A, RS_>, dim3(l.grid), dim3(l.block), 2 * (size_t)C * sizeof(float), s,   \
                    reinterpret_cast<const XT*>(x), dr, reinterpret_cast<const XT*>(res), reinterpret_cast<XT*>(dres), \
                    nvec, C, inv_R, save_mean, save_invstd, gamma, beta, dsums, reinterpret_cast<XT*>(dx), dgamma, \
                    dbeta);                                                                                        \
  } while (0)
#define HD_DX(T, A, XT)               \
  do {                                \
    if (res) HD_DX_R(T, A, 1, XT);    \
    else HD_DX_R(T, A, 0, XT);        \
  } while (0)
  if (dtype == 0) HD_ACT_DISPATCH(float, HD_DX, float);
  else HD_ACT_DISPATCH(__hip_bfloat16, HD_DX, __hip_bfloat16);
#undef HD_DX
#undef HD_DX_R
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_bias_act_fwd(const void* x, int dtype, int64_t R, int C, const float* bias, int act, void* y,
                                     hd_stream stream) {
  hd_status st = common_checks(dtype, R, C);
  if (st != HD_OK) return st;
  if (!x || !bias || !y) return HD_ERR_NULL;
  if (act != ACT_NONE && act != ACT_RELU && act != ACT_SILU) return HD_ERR_SHAPE;
  if (!hd_aligned16(x) || !hd_aligned16(y)) return HD_ERR_ALIGN;
  const int V = dtype == 0 ? 4 : 8;
  const long long nvec = R * (long long)(C / V);
  const Launch l = plan(nvec / 4 + 1, C / V, 2048);
  hipStream_t s = (hipStream_t)stream;
#define HD_BIAS_F(T, A, XT)                                                                                        \
  (void)hd_launch(bias_act_fwd_kernel<XT, A>, dim3(l.grid), dim3(l.block), 0, s, reinterpret_cast<const XT*>(x), nvec, C, \
                  bias, reinterpret_cast<XT*>(y))
  if (dtype == 0) HD_ACT_DISPATCH(float, HD_BIAS_F, float);
  else HD_ACT_DISPATCH(__hip_bfloat16, HD_BIAS_F, __hip_bfloat16);
#undef HD_BIAS_F
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_bias_act_bwd(const void* y, const void* dy, int dtype, int64_t R, int C, int act, void* dx,
                                     float* dsums, hd_stream stream) {
  hd_status st = common_checks(dtype, R, C);
  if (st != HD_OK) return st;
  if (!dy || !dsums) return HD_ERR_NULL;
  if (act != ACT_NONE && act != ACT_RELU) return HD_ERR_SHAPE;  // the output alone decides act' only for these two
  if (act == ACT_RELU && (!y || !dx)) return HD_ERR_NULL;
  if (!hd_aligned16(dy) || (y && !hd_aligned16(y)) || (dx && !hd_aligned16(dx))) return HD_ERR_ALIGN;
  const int V = dtype == 0 ? 4 : 8;
  const long long nvec = R * (long long)(C / V);
  const Launch l = plan(nvec / 8 + 1, C / V, 512);
  const size_t lds = reduce_lds_bytes(l, C / V, C);
  hipStream_t s = (hipStream_t)stream;
#define HD_BIAS_B(XT, A, W)                                                                                        \
  (void)hd_launch(bias_act_bwd_kernel<XT, A, W>, dim3(l.grid), dim3(l.block), lds, s, reinterpret_cast<const XT*>(y), \
                  reinterpret_cast<const XT*>(dy), nvec, C, reinterpret_cast<XT*>(dx), dsums)
  if (dtype == 0) {
    if (act == ACT_RELU) HD_BIAS_B(float, ACT_RELU, true);
    else if (dx) HD_BIAS_B(float, ACT_NONE, true);
    else HD_BIAS_B(float, ACT_NONE, false);
  } else {
    if (act == ACT_RELU) HD_BIAS_B(__hip_bfloat16, ACT_RELU, true);
    else if (dx) HD_BIAS_B(__hip_bfloat16, ACT_NONE, true);
    else HD_BIAS_B(__hip_bfloat16, ACT_NONE, false);
  }
#undef HD_BIAS_B
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" int hd_bn_act_replicas(void) { return kRep; }
