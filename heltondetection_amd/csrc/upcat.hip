// upcat.hip — nearest-neighbour 2x upsampling of one channels-last tensor concatenated with a second one, in one pass:
//     out[b, :, y, x] = [ small[b, :, y / 2, x / 2] | big[b, :, y, x] ]
// the top-down junctions of the YOLOv5 PAFPN neck (reference README.md:9; two per forward).
//
// Why hand-written: PyTorch-ROCm runs the upsample as a float32 kernel (31 us at [16,256,20,20]), and torch.cat of a
// float32 and a bfloat16 tensor promotes the whole concatenation to float32 — the following convolution then casts it
// back (10 us), and the backward repeats the detour (upsample backward, float<->bf16 copies).  Here the forward is one
// gather of 16-byte channel vectors straight into the bf16 concatenation, and the backward sums the four fine-grid
// gradients of every coarse pixel (float32 accumulate); the gradient of `big` is a channel slice of the incoming
// gradient, i.e. a view, not a copy.
#include <hip/hip_bf16.h>

#include "hd_common.h"
#include "hd_vec.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void upcat_fwd_kernel(const T* __restrict__ small, long long ld_s, const T* __restrict__ big,
                                                        long long ld_b, T* __restrict__ out, long long ld_o, int B, int H,
                                                        int W, int Cs, int Cb) {
  // H, W: the FINE grid (big / out); small is [B, H/2, W/2, Cs]
  constexpr int V = RowVec<T>::N;
  const int vs = Cs / V, vt = (Cs + Cb) / V;
  const long long total = (long long)B * H * W * vt;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % vt);
    long long p = i / vt;  // fine pixel
    const int x = (int)(p % W);
    const long long r = p / W;
    const int y = (int)(r % H);
    const int b = (int)(r / H);
    uint4 val;
    if (v < vs) {
      const long long ps = ((long long)b * (H / 2) + y / 2) * (W / 2) + x / 2;
      val = *reinterpret_cast<const uint4*>(small + ps * ld_s + (long long)v * V);
    } else {
      val = *reinterpret_cast<const uint4*>(big + p * ld_b + (long long)(v - vs) * V);
    }
    *reinterpret_cast<uint4*>(out + p * ld_o + (long long)v * V) = val;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void upcat_bwd_kernel(const T* __restrict__ g, long long ld_g, T* __restrict__ d_small,
                                                        long long ld_d, int B, int H, int W, int Cs) {
  // H, W: the FINE grid; d_small[b, y, x, :] = sum of g[b, 2y + {0,1}, 2x + {0,1}, :Cs]
  constexpr int V = RowVec<T>::N;
  const int vs = Cs / V, Hs = H / 2, Ws = W / 2;
  const long long total = (long long)B * Hs * Ws * vs;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % vs);
    const long long p = i / vs;  // coarse pixel
    const int x = (int)(p % Ws);
    const long long r = p / Ws;
    const int y = (int)(r % Hs);
    const int b = (int)(r / Hs);
    const long long f00 = ((long long)b * H + 2 * y) * W + 2 * x;
    float a[4][V], s[V];
    RowVec<T>::load(g + f00 * ld_g + (long long)v * V, a[0]);
    RowVec<T>::load(g + (f00 + 1) * ld_g + (long long)v * V, a[1]);
    RowVec<T>::load(g + (f00 + W) * ld_g + (long long)v * V, a[2]);
    RowVec<T>::load(g + (f00 + W + 1) * ld_g + (long long)v * V, a[3]);
#pragma unroll
    for (int k = 0; k < V; ++k) s[k] = (a[0][k] + a[1][k]) + (a[2][k] + a[3][k]);
    RowVec<T>::store(d_small + p * ld_d + (long long)v * V, s);
  }
}

inline unsigned grid_for(long long total) {
  long long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" hd_status hd_upsample2x_cat_fwd(const void* small, int64_t small_row_stride, const void* big, int64_t big_row_stride,
                                           int dtype, int B, int H, int W, int Cs, int Cb, void* out, int64_t out_row_stride,
                                           hd_stream stream) {
  if (!small || !big || !out) return HD_ERR_NULL;
  const int V = dtype == 0 ? 4 : 8;
  if ((dtype != 0 && dtype != 1) || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || Cs < V || Cb < V || Cs % V || Cb % V)
    return HD_ERR_SHAPE;
  if (small_row_stride == 0) small_row_stride = Cs;
  if (big_row_stride == 0) big_row_stride = Cb;
  if (out_row_stride == 0) out_row_stride = Cs + Cb;
  if (small_row_stride < Cs || big_row_stride < Cb || out_row_stride < Cs + Cb || small_row_stride % V || big_row_stride % V ||
      out_row_stride % V)
    return HD_ERR_SHAPE;
  if (!hd_aligned16(small) || !hd_aligned16(big) || !hd_aligned16(out)) return HD_ERR_ALIGN;
  const long long total = (long long)B * H * W * ((Cs + Cb) / V);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == 0)
    (void)hd_launch(upcat_fwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)small, small_row_stride,
                    (const float*)big, big_row_stride, (float*)out, out_row_stride, B, H, W, Cs, Cb);
  else
    (void)hd_launch(upcat_fwd_kernel<__hip_bfloat16>, dim3(grid_for(total)), dim3(256), 0, s, (const __hip_bfloat16*)small,
                    small_row_stride, (const __hip_bfloat16*)big, big_row_stride, (__hip_bfloat16*)out, out_row_stride, B, H, W,
                    Cs, Cb);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_upsample2x_cat_bwd(const void* grad_out, int64_t grad_row_stride, int dtype, int B, int H, int W, int Cs,
                                           void* d_small, int64_t d_small_row_stride, hd_stream stream) {
  if (!grad_out || !d_small) return HD_ERR_NULL;
  const int V = dtype == 0 ? 4 : 8;
  if ((dtype != 0 && dtype != 1) || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || Cs < V || Cs % V) return HD_ERR_SHAPE;
  if (d_small_row_stride == 0) d_small_row_stride = Cs;
  if (grad_row_stride < Cs || d_small_row_stride < Cs || grad_row_stride % V || d_small_row_stride % V) return HD_ERR_SHAPE;
  if (!hd_aligned16(grad_out) || !hd_aligned16(d_small)) return HD_ERR_ALIGN;
  const long long total = (long long)B * (H / 2) * (W / 2) * (Cs / V);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == 0)
    (void)hd_launch(upcat_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)grad_out, grad_row_stride,
                    (float*)d_small, d_small_row_stride, B, H, W, Cs);
  else
    (void)hd_launch(upcat_bwd_kernel<__hip_bfloat16>, dim3(grid_for(total)), dim3(256), 0, s, (const __hip_bfloat16*)grad_out,
                    grad_row_stride, (__hip_bfloat16*)d_small, d_small_row_stride, B, H, W, Cs);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}
