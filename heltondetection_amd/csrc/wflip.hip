// wflip.hip — the weights of the 3x3 data gradients, for ALL layers of a model in one launch.
//
// hd_conv3x3_fwd is also the data gradient of a 3x3 / stride 1 convolution when it is given the flipped, transposed
// weights  wt[c][ky][kx][n] = w[n][2 - ky][2 - kx][c]  (include/hd_ops.h).  Built per layer with tensor ops that is a flip
// kernel + a permuting copy per layer and step: 20 launches of ~5-8 us in the YOLOv5s step for 3.7 MB of weights.  The
// weights only change in the optimizer step, so one launch at the head of the step rebuilds every layer's copy: a batch of
// 9 x layers tap-reversed [Cout, Cin] -> [Cin, Cout] transposes through 32 x 32 LDS tiles (coalesced on both sides).
#include "hd_common.h"

namespace {

struct FlipBatch {
  int n;                                 // layers
  const unsigned short* src[HD_MAX_FLIP];  // w  [Cout][3][3][Cin] bf16 (channels-last memory of [Cout, Cin, 3, 3])
  unsigned short* dst[HD_MAX_FLIP];        // wt [Cin][3][3][Cout]
  int cout[HD_MAX_FLIP], cin[HD_MAX_FLIP], taps[HD_MAX_FLIP];   // taps: 9 (3x3) or 1 (1x1: a plain transpose)
  int tile0[HD_MAX_FLIP + 1];            // first 32 x 32 tile of layer l (tiles = taps x ceil(Cout/32) x ceil(Cin/32))
};

__global__ __launch_bounds__(256) void conv3x3_wflip_kernel(const FlipBatch fb) {
  __shared__ unsigned short tile[32][33];
  int l = 0;
  while (l + 1 < fb.n && (int)blockIdx.x >= fb.tile0[l + 1]) ++l;
  const int N = fb.cout[l], C = fb.cin[l], T = fb.taps[l];
  const int tn = (N + 31) / 32, tc = (C + 31) / 32;
  int t = blockIdx.x - fb.tile0[l];
  const int tap = t / (tn * tc);   // < T
  t -= tap * tn * tc;
  const int n0 = (t / tc) * 32, c0 = (t % tc) * 32;
  const unsigned short* __restrict__ w = fb.src[l];
  unsigned short* __restrict__ wt = fb.dst[l];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int r = ty; r < 32; r += 8) {  // rows n, columns c (contiguous in w)
    const int n = n0 + r, c = c0 + tx;
    tile[r][tx] = (n < N && c < C) ? w[((long long)n * T + tap) * C + c] : (unsigned short)0;
  }
  __syncthreads();
  const int rtap = T - 1 - tap;  // (2 - ky, 2 - kx)
#pragma unroll
  for (int r = ty; r < 32; r += 8) {  // rows c, columns n (contiguous in wt)
    const int c = c0 + r, n = n0 + tx;
    if (c < C && n < N) wt[((long long)c * T + rtap) * N + n] = tile[tx][r];
  }
}

}  // namespace

extern "C" hd_status hd_conv_flip_weights(int n_layers, const void* const* w, void* const* wt, const int* cout, const int* cin,
                                          const int* taps, hd_stream stream) {
  if (n_layers < 0 || n_layers > HD_MAX_FLIP) return HD_ERR_SHAPE;
  if (n_layers == 0) return HD_OK;
  if (!w || !wt || !cout || !cin) return HD_ERR_NULL;
  FlipBatch fb;
  fb.n = n_layers;
  int tiles = 0;
  for (int l = 0; l < n_layers; ++l) {
    if (!w[l] || !wt[l]) return HD_ERR_NULL;
    const int t = taps ? taps[l] : 9;
    if (cout[l] <= 0 || cin[l] <= 0 || cout[l] > 8192 || cin[l] > 8192 || (t != 1 && t != 9)) return HD_ERR_SHAPE;
    fb.src[l] = static_cast<const unsigned short*>(w[l]);
    fb.dst[l] = static_cast<unsigned short*>(wt[l]);
    fb.cout[l] = cout[l];
    fb.cin[l] = cin[l];
    fb.taps[l] = t;
    fb.tile0[l] = tiles;
    tiles += t * ((cout[l] + 31) / 32) * ((cin[l] + 31) / 32);
  }
  for (int l = n_layers; l <= HD_MAX_FLIP; ++l) fb.tile0[l] = tiles;
  for (int l = n_layers; l < HD_MAX_FLIP; ++l) fb.taps[l] = 9;
  (void)hd_launch(conv3x3_wflip_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, fb);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_conv3x3_flip_weights(int n_layers, const void* const* w, void* const* wt, const int* cout, const int* cin,
                                             hd_stream stream) {
  return hd_conv_flip_weights(n_layers, w, wt, cout, cin, nullptr, stream);
}
