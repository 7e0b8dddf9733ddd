// anchors.hip — anchor grid generation (FasterRCNN RPN, reference README.md:8).
//
// HBM-write bound: 16 bytes per anchor out, nothing in (the per-level base anchors travel in
// the kernel argument block).  One lane per anchor, one 16-byte store per lane: a wave writes
// 1 KiB contiguous.  Order is level-major, then y, x, anchor; the float
// arithmetic is the oracle's orc_anchor_grid() so the result is bit-identical.
#include <math.h>

#include "hd_common.h"

namespace {

struct AnchorLevels {
  int n_levels;
  int A;
  float offset;
  int h[HD_MAX_LEVELS], w[HD_MAX_LEVELS];
  float stride[HD_MAX_LEVELS];
  long long first[HD_MAX_LEVELS + 1];  // first anchor index of each level
  float base[HD_MAX_LEVELS][HD_MAX_ANCHORS][4];
};

__global__ __launch_bounds__(256) void anchor_grid_kernel(const AnchorLevels lv, float4* __restrict__ out) {
  const long long total = lv.first[lv.n_levels];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < HD_MAX_LEVELS; ++k)
      if (k < lv.n_levels && i >= lv.first[k]) l = k;
    const long long r = i - lv.first[l];
    const int a = (int)(r % lv.A);
    const long long cell = r / lv.A;
    const int x = (int)(cell % lv.w[l]);
    const int y = (int)(cell / lv.w[l]);
    const float sx = ((float)x + lv.offset) * lv.stride[l];
    const float sy = ((float)y + lv.offset) * lv.stride[l];
    out[i] = make_float4(lv.base[l][a][0] + sx, lv.base[l][a][1] + sy, lv.base[l][a][2] + sx,
                         lv.base[l][a][3] + sy);
  }
}

}  // namespace

extern "C" hd_status hd_anchor_base(const float* sizes, int n_sizes, const float* ratios, int n_ratios,
                                    float* out_base) {
  if (!sizes || !ratios || !out_base) return HD_ERR_NULL;
  if (n_sizes <= 0 || n_ratios <= 0) return HD_ERR_SHAPE;
  int a = 0;
  for (int r = 0; r < n_ratios; ++r) {
    const float hr = sqrtf(ratios[r]);
    for (int s = 0; s < n_sizes; ++s, ++a) {
      const float w = sizes[s] / hr;
      const float h = sizes[s] * hr;
      out_base[a * 4 + 0] = -w * 0.5f;
      out_base[a * 4 + 1] = -h * 0.5f;
      out_base[a * 4 + 2] = w * 0.5f;
      out_base[a * 4 + 3] = h * 0.5f;
    }
  }
  return HD_OK;
}

extern "C" hd_status hd_anchor_grid(int n_levels, const int* feat_h, const int* feat_w, const float* strides,
                                    const float* base_anchors, int A, float offset, float* out,
                                    hd_stream stream) {
  if (!feat_h || !feat_w || !strides || !base_anchors || !out) return HD_ERR_NULL;
  if (n_levels <= 0 || n_levels > HD_MAX_LEVELS || A <= 0 || A > HD_MAX_ANCHORS) return HD_ERR_SHAPE;
  if (!hd_aligned16(out)) return HD_ERR_ALIGN;
  AnchorLevels lv;
  lv.n_levels = n_levels;
  lv.A = A;
  lv.offset = offset;
  long long first = 0;
  for (int l = 0; l < n_levels; ++l) {
    if (feat_h[l] <= 0 || feat_w[l] <= 0) return HD_ERR_SHAPE;
    lv.h[l] = feat_h[l];
    lv.w[l] = feat_w[l];
    lv.stride[l] = strides[l];
    lv.first[l] = first;
    first += (long long)feat_h[l] * feat_w[l] * A;
    for (int a = 0; a < A; ++a)
      for (int k = 0; k < 4; ++k) lv.base[l][a][k] = base_anchors[((size_t)l * A + a) * 4 + k];
  }
  for (int l = n_levels; l <= HD_MAX_LEVELS; ++l) lv.first[l] = first;
  long long blocks = (first + 255) / 256;
  if (blocks > 2048) blocks = 2048;  // grid-stride beyond 8 workgroups per CU
  hipLaunchKernelGGL(anchor_grid_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, lv,
                     reinterpret_cast<float4*>(out));
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}
