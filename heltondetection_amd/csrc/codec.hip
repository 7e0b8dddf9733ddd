// codec.hip — box <-> delta codec and the CIoU of matched pairs (fwd + hand-derived bwd).
//
// All HBM-streaming, one lane per box, 16-byte loads and stores ("HBM, tiny").
// The CIoU arithmetic mirrors oracle/hd_oracle.c:ciou_xyxy() term by term; atanf/expf/logf are
// device libm, so these are tolerance-checked (1e-4 rel), not bit-checked.
#include "hd_common.h"
#include "hd_ciou.h"

namespace {



__global__ __launch_bounds__(256) void box_encode_kernel(const float4* __restrict__ src,
                                                         const float4* __restrict__ dst, long long n,
                                                         float4* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float4 s = src[i], d = dst[i];
    const float sw = s.z - s.x, sh = s.w - s.y;
    const float scx = s.x + 0.5f * sw, scy = s.y + 0.5f * sh;
    const float dw = d.z - d.x, dh = d.w - d.y;
    const float dcx = d.x + 0.5f * dw, dcy = d.y + 0.5f * dh;
    out[i] = make_float4((dcx - scx) / sw, (dcy - scy) / sh, logf(dw / sw), logf(dh / sh));
  }
}

__global__ __launch_bounds__(256) void box_decode_kernel(const float4* __restrict__ anchors,
                                                         const float4* __restrict__ deltas, long long n,
                                                         float img_h, float img_w, float4* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    out[i] = hd_delta_decode(anchors[i], deltas[i], img_h, img_w);
}

__global__ __launch_bounds__(256) void ciou_fwd_kernel(const float4* __restrict__ p, const float4* __restrict__ t,
                                                       long long n, float* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    out[i] = hd_ciou<false>(p[i], t[i], nullptr);
}

__global__ __launch_bounds__(256) void ciou_bwd_kernel(const float4* __restrict__ p, const float4* __restrict__ t,
                                                       const float* __restrict__ go, long long n,
                                                       float4* __restrict__ gp) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float g[4];
    hd_ciou<true>(p[i], t[i], g);
    const float s = go[i];
    gp[i] = make_float4(s * g[0], s * g[1], s * g[2], s * g[3]);
  }
}

inline unsigned stream_grid(long long n) {
  long long b = (n + 255) / 256;
  return (unsigned)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" hd_status hd_box_encode(const float* src, const float* dst, int64_t n, float* out_deltas,
                                   hd_stream stream) {
  if (n < 0) return HD_ERR_SHAPE;
  if (n == 0) return HD_OK;
  if (!src || !dst || !out_deltas) return HD_ERR_NULL;
  if (!hd_aligned16(src) || !hd_aligned16(dst) || !hd_aligned16(out_deltas)) return HD_ERR_ALIGN;
  hipLaunchKernelGGL(box_encode_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(src), reinterpret_cast<const float4*>(dst), (long long)n,
                     reinterpret_cast<float4*>(out_deltas));
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_box_decode(const float* anchors, const float* deltas, int64_t n, float img_h,
                                   float img_w, float* out_boxes, hd_stream stream) {
  if (n < 0) return HD_ERR_SHAPE;
  if (n == 0) return HD_OK;
  if (!anchors || !deltas || !out_boxes) return HD_ERR_NULL;
  if (!hd_aligned16(anchors) || !hd_aligned16(deltas) || !hd_aligned16(out_boxes)) return HD_ERR_ALIGN;
  hipLaunchKernelGGL(box_decode_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(anchors), reinterpret_cast<const float4*>(deltas),
                     (long long)n, img_h, img_w, reinterpret_cast<float4*>(out_boxes));
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_ciou_fwd(const float* pred, const float* target, int64_t P, float* ciou,
                                 hd_stream stream) {
  if (P < 0) return HD_ERR_SHAPE;
  if (P == 0) return HD_OK;
  if (!pred || !target || !ciou) return HD_ERR_NULL;
  if (!hd_aligned16(pred) || !hd_aligned16(target)) return HD_ERR_ALIGN;
  hipLaunchKernelGGL(ciou_fwd_kernel, dim3(stream_grid(P)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(pred), reinterpret_cast<const float4*>(target), (long long)P,
                     ciou);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_ciou_bwd(const float* pred, const float* target, const float* grad_ciou, int64_t P,
                                 float* grad_pred, hd_stream stream) {
  if (P < 0) return HD_ERR_SHAPE;
  if (P == 0) return HD_OK;
  if (!pred || !target || !grad_ciou || !grad_pred) return HD_ERR_NULL;
  if (!hd_aligned16(pred) || !hd_aligned16(target) || !hd_aligned16(grad_pred)) return HD_ERR_ALIGN;
  hipLaunchKernelGGL(ciou_bwd_kernel, dim3(stream_grid(P)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(pred), reinterpret_cast<const float4*>(target), grad_ciou,
                     (long long)P, reinterpret_cast<float4*>(grad_pred));
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}
