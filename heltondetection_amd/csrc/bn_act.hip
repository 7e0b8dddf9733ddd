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

#include "hd_common.h"

namespace {

template <typename T>
struct RowVec;
template <>
struct RowVec<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float* v) {
    const float4 r = *reinterpret_cast<const float4*>(p);
    v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
  }
  static __device__ __forceinline__ void store(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <>
struct RowVec<__hip_bfloat16> {
  static constexpr int N = 8;
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
      const __hip_bfloat16 lo = __float2bfloat16(v[2 * i]), hi = __float2bfloat16(v[2 * i + 1]);
      w[i] = (uint32_t)(*reinterpret_cast<const uint16_t*>(&lo)) |
             ((uint32_t)(*reinterpret_cast<const uint16_t*>(&hi)) << 16);
    }
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

enum { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2 };

template <int ACT>
__device__ __forceinline__ float act_fwd(float z) {
  if (ACT == ACT_SILU) return z / (1.0f + __expf(-z));
  if (ACT == ACT_RELU) return z > 0.0f ? z : 0.0f;
  return z;
}
// d act(z) / dz
template <int ACT>
__device__ __forceinline__ float act_grad(float z) {
  if (ACT == ACT_SILU) {
    const float s = 1.0f / (1.0f + __expf(-z));
    return s * (1.0f + z * (1.0f - s));
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

// ---- forward statistics -------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, long long nvec, int C,
                                                       float* __restrict__ sums /* [kRep, 2, C] */) {
  constexpr int V = RowVec<T>::N;
  extern __shared__ float s_acc[];  // [2*C]
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) s_acc[i] = 0.0f;
  __syncthreads();
  const int cg = C / V;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = (int)(i0 % cg);
  float s[V], ss[V];
#pragma unroll
  for (int k = 0; k < V; ++k) s[k] = ss[k] = 0.0f;
  long long i = i0;
  for (; i + 3 * stride < nvec; i += 4 * stride) {  // 4 independent 16-byte loads in flight per lane
    float v0[V], v1[V], v2[V], v3[V];
    RowVec<T>::load(x + i * V, v0);
    RowVec<T>::load(x + (i + stride) * V, v1);
    RowVec<T>::load(x + (i + 2 * stride) * V, v2);
    RowVec<T>::load(x + (i + 3 * stride) * V, v3);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      s[k] += (v0[k] + v1[k]) + (v2[k] + v3[k]);
      ss[k] += (v0[k] * v0[k] + v1[k] * v1[k]) + (v2[k] * v2[k] + v3[k] * v3[k]);
    }
  }
  for (; i < nvec; i += stride) {
    float v[V];
    RowVec<T>::load(x + i * V, v);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      s[k] += v[k];
      ss[k] += v[k] * v[k];
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) {
    atomicAdd(&s_acc[g * V + k], s[k]);
    atomicAdd(&s_acc[C + g * V + k], ss[k]);
  }
  __syncthreads();
  float* dst = sums + (size_t)(blockIdx.x % kRep) * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) atomicAdd(&dst[i], s_acc[i]);
}

__device__ __forceinline__ float rep_sum(const float* __restrict__ p, int idx, int C) {
  float v = 0.0f;
#pragma unroll
  for (int r = 0; r < kRep; ++r) v += p[(size_t)r * 2 * C + idx];
  return v;
}

// ---- forward apply ------------------------------------------------------------------------------
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, long long nvec, int C, float inv_R,
                                                       float unbias, const float* __restrict__ sums,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, float momentum,
                                                       float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, float* __restrict__ save_mean,
                                                       float* __restrict__ save_invstd, T* __restrict__ y) {
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
  for (long long i = i0; i < nvec; i += stride) {
    float v[V];
    RowVec<T>::load(x + i * V, v);
#pragma unroll
    for (int k = 0; k < V; ++k) v[k] = act_fwd<ACT>(a[k] * v[k] + b[k]);
    RowVec<T>::store(y + i * V, v);
  }
}

// inference: running statistics, no reduction
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_infer_kernel(const T* __restrict__ x, long long nvec, int C,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       const float* __restrict__ running_mean,
                                                       const float* __restrict__ running_var, float eps,
                                                       T* __restrict__ y) {
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
    for (int k = 0; k < V; ++k) v[k] = act_fwd<ACT>(a[k] * v[k] + b[k]);
    RowVec<T>::store(y + i * V, v);
  }
}

// ---- backward reductions ------------------------------------------------------------------------
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                            long long nvec, int C,
                                                            const float* __restrict__ save_mean,
                                                            const float* __restrict__ save_invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            float* __restrict__ dsums /* [kRep,2,C]: dbeta, dgamma */) {
  constexpr int V = RowVec<T>::N;
  extern __shared__ float s_acc[];
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) s_acc[i] = 0.0f;
  __syncthreads();
  const int cg = C / V;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = (int)(i0 % cg);
  float mu[V], is[V], ga[V], be[V], s1[V], s2[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const int c = g * V + k;
    mu[k] = save_mean[c];
    is[k] = save_invstd[c];
    ga[k] = gamma[c];
    be[k] = beta[c];
    s1[k] = s2[k] = 0.0f;
  }
  long long i = i0;
  for (; i + stride < nvec; i += 2 * stride) {  // 4 independent 16-byte loads in flight per lane
    float xv[V], gv[V], xw[V], gw[V];
    RowVec<T>::load(x + i * V, xv);
    RowVec<T>::load(dy + i * V, gv);
    RowVec<T>::load(x + (i + stride) * V, xw);
    RowVec<T>::load(dy + (i + stride) * V, gw);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float xh = (xv[k] - mu[k]) * is[k];
      const float dz = gv[k] * act_grad<ACT>(ga[k] * xh + be[k]);
      const float xh2 = (xw[k] - mu[k]) * is[k];
      const float dz2 = gw[k] * act_grad<ACT>(ga[k] * xh2 + be[k]);
      s1[k] += dz + dz2;
      s2[k] += dz * xh + dz2 * xh2;
    }
  }
  for (; i < nvec; i += stride) {
    float xv[V], gv[V];
    RowVec<T>::load(x + i * V, xv);
    RowVec<T>::load(dy + i * V, gv);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float xh = (xv[k] - mu[k]) * is[k];
      const float dz = gv[k] * act_grad<ACT>(ga[k] * xh + be[k]);
      s1[k] += dz;
      s2[k] += dz * xh;
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) {
    atomicAdd(&s_acc[g * V + k], s1[k]);
    atomicAdd(&s_acc[C + g * V + k], s2[k]);
  }
  __syncthreads();
  float* dst = dsums + (size_t)(blockIdx.x % kRep) * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) atomicAdd(&dst[i], s_acc[i]);
}

template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const T* __restrict__ x, const T* __restrict__ dy,
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
  float mu[V], is[V], ga[V], be[V], k1[V], k2[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const int c = g * V + k;
    mu[k] = save_mean[c];
    is[k] = save_invstd[c];
    ga[k] = gamma[c];
    be[k] = beta[c];
    k1[k] = s_coef[c];
    k2[k] = s_coef[C + c];
  }
  for (long long i = i0; i < nvec; i += stride) {
    float xv[V], gv[V];
    RowVec<T>::load(x + i * V, xv);
    RowVec<T>::load(dy + i * V, gv);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float xh = (xv[k] - mu[k]) * is[k];
      const float dz = gv[k] * act_grad<ACT>(ga[k] * xh + be[k]);
      xv[k] = ga[k] * is[k] * (dz - k1[k] - xh * k2[k]);
    }
    RowVec<T>::store(dx + i * V, xv);
  }
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

extern "C" hd_status hd_bn_act_stats(const void* x, int dtype, int64_t R, int C, float* sums, hd_stream stream) {
  hd_status st = common_checks(dtype, R, C);
  if (st != HD_OK) return st;
  if (!x || !sums) return HD_ERR_NULL;
  if (!hd_aligned16(x)) return HD_ERR_ALIGN;
  const int V = dtype == 0 ? 4 : 8;
  const long long nvec = R * (long long)(C / V);
  const Launch l = plan(nvec / 8 + 1, C / V, 512);  // ~8 vectors per lane: few workgroups -> few atomics
  const size_t lds = 2 * (size_t)C * sizeof(float);
  if (dtype == 0)
    hipLaunchKernelGGL((bn_stats_kernel<float>), dim3(l.grid), dim3(l.block), lds, (hipStream_t)stream,
                       reinterpret_cast<const float*>(x), nvec, C, sums);
  else
    hipLaunchKernelGGL((bn_stats_kernel<__hip_bfloat16>), dim3(l.grid), dim3(l.block), lds, (hipStream_t)stream,
                       reinterpret_cast<const __hip_bfloat16*>(x), nvec, C, sums);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_bn_act_apply(const void* x, int dtype, int64_t R, int C, const float* sums,
                                     const float* gamma, const float* beta, float eps, float momentum,
                                     float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                                     int act, void* y, hd_stream stream) {
  hd_status st = common_checks(dtype, R, C);
  if (st != HD_OK) return st;
  if (!x || !sums || !gamma || !beta || !save_mean || !save_invstd || !y) return HD_ERR_NULL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return HD_ERR_NULL;
  if (!hd_aligned16(x) || !hd_aligned16(y)) return HD_ERR_ALIGN;
  const int V = dtype == 0 ? 4 : 8;
  const long long nvec = R * (long long)(C / V);
  const Launch l = plan(nvec / 4 + 1, C / V, 2048);  // >= 4 vectors per lane: the per-workgroup prologue is not free
  const float inv_R = 1.0f / (float)R;
  const float unbias = R > 1 ? (float)R / (float)(R - 1) : 1.0f;
  hipStream_t s = (hipStream_t)stream;
#define HD_APPLY(T, A, XT)                                                                                         \
  hipLaunchKernelGGL((bn_apply_kernel<XT, A>), dim3(l.grid), dim3(l.block), 2 * (size_t)C * sizeof(float), s, reinterpret_cast<const XT*>(x), \
                     nvec, C, inv_R, unbias, sums, gamma, beta, eps, momentum, running_mean, running_var,        \
                     save_mean, save_invstd, reinterpret_cast<XT*>(y))
  if (dtype == 0) HD_ACT_DISPATCH(float, HD_APPLY, float);
  else HD_ACT_DISPATCH(__hip_bfloat16, HD_APPLY, __hip_bfloat16);
#undef HD_APPLY
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_bn_act_infer(const void* x, int dtype, int64_t R, int C, const float* gamma,
                                     const float* beta, const float* running_mean, const float* running_var,
                                     float eps, int act, void* y, hd_stream stream) {
  hd_status st = common_checks(dtype, R, C);
  if (st != HD_OK) return st;
  if (!x || !gamma || !beta || !running_mean || !running_var || !y) return HD_ERR_NULL;
  if (!hd_aligned16(x) || !hd_aligned16(y)) return HD_ERR_ALIGN;
  const int V = dtype == 0 ? 4 : 8;
  const long long nvec = R * (long long)(C / V);
  const Launch l = plan(nvec / 4 + 1, C / V, 2048);  // >= 4 vectors per lane: the per-workgroup prologue is not free
  hipStream_t s = (hipStream_t)stream;
#define HD_INFER(T, A, XT)                                                                                         \
  hipLaunchKernelGGL((bn_infer_kernel<XT, A>), dim3(l.grid), dim3(l.block), 0, s, reinterpret_cast<const XT*>(x), \
                     nvec, C, gamma, beta, running_mean, running_var, eps, reinterpret_cast<XT*>(y))
  if (dtype == 0) HD_ACT_DISPATCH(float, HD_INFER, float);
  else HD_ACT_DISPATCH(__hip_bfloat16, HD_INFER, __hip_bfloat16);
#undef HD_INFER
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_bn_act_bwd_reduce(const void* x, const void* dy, int dtype, int64_t R, int C,
                                          const float* save_mean, const float* save_invstd, const float* gamma,
                                          const float* beta, int act, float* dsums, hd_stream stream) {
  hd_status st = common_checks(dtype, R, C);
  if (st != HD_OK) return st;
  if (!x || !dy || !save_mean || !save_invstd || !gamma || !beta || !dsums) return HD_ERR_NULL;
  if (!hd_aligned16(x) || !hd_aligned16(dy)) return HD_ERR_ALIGN;
  const int V = dtype == 0 ? 4 : 8;
  const long long nvec = R * (long long)(C / V);
  const Launch l = plan(nvec / 8 + 1, C / V, 512);  // ~8 vectors per lane: few workgroups -> few atomics
  const size_t lds = 2 * (size_t)C * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
#define HD_RED(T, A, XT)                                                                                          \
  hipLaunchKernelGGL((bn_bwd_reduce_kernel<XT, A>), dim3(l.grid), dim3(l.block), lds, s,                         \
                     reinterpret_cast<const XT*>(x), reinterpret_cast<const XT*>(dy), nvec, C, save_mean,        \
                     save_invstd, gamma, beta, dsums)
  if (dtype == 0) HD_ACT_DISPATCH(float, HD_RED, float);
  else HD_ACT_DISPATCH(__hip_bfloat16, HD_RED, __hip_bfloat16);
#undef HD_RED
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_bn_act_bwd_dx(const void* x, const void* dy, int dtype, int64_t R, int C,
                                      const float* save_mean, const float* save_invstd, const float* gamma,
                                      const float* beta, int act, const float* dsums, void* dx, float* dgamma,
                                      float* dbeta, hd_stream stream) {
  hd_status st = common_checks(dtype, R, C);
  if (st != HD_OK) return st;
  if (!x || !dy || !save_mean || !save_invstd || !gamma || !beta || !dsums || !dx || !dgamma || !dbeta)
    return HD_ERR_NULL;
  if (!hd_aligned16(x) || !hd_aligned16(dy) || !hd_aligned16(dx)) return HD_ERR_ALIGN;
  const int V = dtype == 0 ? 4 : 8;
  const long long nvec = R * (long long)(C / V);
  const Launch l = plan(nvec / 4 + 1, C / V, 2048);  // >= 4 vectors per lane: the per-workgroup prologue is not free
  const float inv_R = 1.0f / (float)R;
  hipStream_t s = (hipStream_t)stream;
#define HD_DX(T, A, XT)                                                                                           \
  hipLaunchKernelGGL((bn_bwd_dx_kernel<XT, A>), dim3(l.grid), dim3(l.block), 2 * (size_t)C * sizeof(float), s, reinterpret_cast<const XT*>(x), \
                     reinterpret_cast<const XT*>(dy), nvec, C, inv_R, save_mean, save_invstd, gamma, beta, dsums,  \
                     reinterpret_cast<XT*>(dx), dgamma, dbeta)
  if (dtype == 0) HD_ACT_DISPATCH(float, HD_DX, float);
  else HD_ACT_DISPATCH(__hip_bfloat16, HD_DX, __hip_bfloat16);
#undef HD_DX
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" int hd_bn_act_replicas(void) { return kRep; }
