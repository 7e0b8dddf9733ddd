// conv3x3s2_wgrad.hip — weight gradient of the FIRST stride-2 3x3 convolution of the YOLOv5 backbone (32 -> 64 channels, 320^2 ->
// 160^2 at 640^2): hd_conv3x3s2_wgrad_c32.
//
//     dw[n][ky][kx][c] = sum over (b, i, j) of dy[b, i, j, n] * x[b, 2 i - 1 + ky, 2 j - 1 + kx, c]
// 15 GFLOP over 157 MB of operands and an 18 432-element result: memory-bound, and the library took 84 us + a zero-fill and a cast.
// hd_conv3x3_wgrad's row kernels need 64-channel tiles on both sides and do not take this layer.  Same walk as the layer's forward
// (conv3x3s2_fwd.hip):
//   * a workgroup of 6 waves walks down `rows_per_wg` output rows of one image.  LDS: the three input rows of the current output
//     row as even / odd pixel planes (64-byte entries: tap kx of consecutive output pixels = consecutive entries of one plane;
//     odd[0] is the zero pixel left of the image) and a ring of two dy rows (pixel stride 144 bytes); the next output row's
//     operands (two x rows + one dy row, 60 KB) travel in registers meanwhile;
//   * wave (n-tile, ky) owns the three 32 x 32 tiles dw[n-tile][ky][kx = 0..2][c] for the whole walk (48 accumulator registers):
//     the reduction runs over the output positions, i.e. over the ROWS of both LDS images, so the fragments are LDS transpose
//     reads (ds_read_b64_tr_b16), 16 positions per k-step, 3 MFMAs per k-step and wave;
//   * at the end of the walk the tiles go to a float32 slab [workgroup][64][9][32] with plain stores, and a second small launch
//     sums the slabs into the bf16 gradient (no zero-fill, no atomics, no cast launch).
#include "hd_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int CI = 32, CO = 64, WOUT = 160, WIN = 2 * WOUT;
constexpr int NWAVE = 6, NTHREAD = NWAVE * 64;
constexpr int EN = CI * 2;                          // bytes per pixel entry of an x plane (64-byte rows: transpose reads need no padding)
constexpr int ODD_OFF = WOUT * EN;                  // even plane: 160 entries; odd plane: 161 (entry 0 = pixel -1 = zero)
constexpr int XROW_BYTES = ODD_OFF + (WOUT + 1) * EN;
constexpr int PXD = CO * 2 + 16;                    // bytes per dy pixel
constexpr int DROW_BYTES = WOUT * PXD;
constexpr int LDS_BYTES = 3 * XROW_BYTES + 2 * DROW_BYTES;
constexpr int XCHUNKS = WIN * CI / 8, DCHUNKS = WOUT * CO / 8;   // 16-byte chunks per x row (1280) / dy row (1280)
constexpr int LD_IT = (2 * XCHUNKS + DCHUNKS) / NTHREAD;         // two x rows + one dy row per fetch (10)
static_assert((2 * XCHUNKS + DCHUNKS) % NTHREAD == 0, "fetch chunks per thread");
constexpr int DW_ELEMS = CO * 9 * CI;               // 18 432

// A fragment of a [position][channel] LDS image for a reduction over positions: lane l gets channel col0 + (l & 31) at the 8
// positions m_base + 8 (l >> 5) .. + 7 (two transpose reads of 4 positions each) — as in pointwise_conv.hip.
__device__ __forceinline__ bf16x8 tr_fragment(const unsigned char* tile, int stride, int col0, int m_base, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const unsigned char* a = tile + (m_base + 8 * (g >> 1) + q) * stride + (col0 + 16 * (g & 1) + 4 * p) * 2;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * stride));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(NTHREAD) void conv3x3s2_wgrad_c32_kernel(const unsigned short* __restrict__ dy, long long ld_dy,
                                                                       const unsigned short* __restrict__ x, long long ld_x,
                                                                       float* __restrict__ slab, int H, int rows_per_wg,
                                                                       int per_img) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [3 x rows as planes | 2 dy rows]
  unsigned char* s_x = smem;
  unsigned char* s_d = smem + 3 * XROW_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / per_img;
  const int r0 = (blockIdx.x % per_img) * rows_per_wg;
  const int r1 = r0 + rows_per_wg < H ? r0 + rows_per_wg : H;  // output rows [r0, r1)
  const int HIN = 2 * H;
  const unsigned short* xb = x + (long long)b * HIN * WIN * ld_x;
  const unsigned short* db = dy + (long long)b * H * WOUT * ld_dy;

  // one fetch = input rows xrow0, xrow0 + 1 and dy row drow: chunk ids [0, 2 XCHUNKS) are x, the rest dy
  u32x4 pre[LD_IT];
  auto gload = [&](u32x4 (&pre)[LD_IT], int xrow0, int drow, bool first_x_only) {
#pragma unroll
    for (int p = 0; p < LD_IT; ++p) {
      const int id = p * NTHREAD + tid;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (id < 2 * XCHUNKS) {
        const int row = xrow0 + id / XCHUNKS, cid = id % XCHUNKS;
        if (row >= 0 && row < HIN && !(first_x_only && id >= XCHUNKS)) v = *reinterpret_cast<const u32x4*>(xb + ((long long)row * WIN + (cid >> 2)) * ld_x + (cid & 3) * 8);
      } else if (!first_x_only) {
        const int cid = id - 2 * XCHUNKS;
        if (drow < H) v = *reinterpret_cast<const u32x4*>(db + ((long long)drow * WOUT + (cid >> 3)) * ld_dy + (cid & 7) * 8);
      }
      pre[p] = v;
    }
  };
  auto xslot = [&](int row) { return s_x + ((row + 3) % 3) * XROW_BYTES; };
  auto lstore = [&](const u32x4 (&pre)[LD_IT], int xrow0, int drow, bool second_x, bool with_dy) {
#pragma unroll
    for (int p = 0; p < LD_IT; ++p) {
      const int id = p * NTHREAD + tid;
      if (id < 2 * XCHUNKS) {
        const int sel = id / XCHUNKS, cid = id % XCHUNKS, px = cid >> 2, c = cid & 3;
        if (sel == 0 || second_x)
          *reinterpret_cast<u32x4*>(xslot(xrow0 + sel) + ((px & 1) ? ODD_OFF + ((px + 1) >> 1) * EN : (px >> 1) * EN) + c * 16) = pre[p];
      } else if (with_dy) {
        const int cid = id - 2 * XCHUNKS;
        *reinterpret_cast<u32x4*>(s_d + (drow & 1) * DROW_BYTES + (cid >> 3) * PXD + (cid & 7) * 16) = pre[p];
      }
    }
  };

  if (tid < 3 * 4)  // odd-plane entry 0 of each x ring row: the zero pixel left of the image (never overwritten)
    *reinterpret_cast<u32x4*>(s_x + (tid >> 2) * XROW_BYTES + ODD_OFF + (tid & 3) * 16) = u32x4{0u, 0u, 0u, 0u};
  u32x4 pre_b[LD_IT];
  gload(pre, 2 * r0 - 1, r0, false);   // x rows 2 r0 - 1, 2 r0; dy row r0
  gload(pre_b, 2 * r0 + 1, 0, true);   // x row 2 r0 + 1 completes output row r0 (only the first row of this fetch is loaded);
                                       // both fetches are in flight before the first LDS store
  lstore(pre, 2 * r0 - 1, r0, true, true);
  lstore(pre_b, 2 * r0 + 1, 0, false, false);
  if (r0 + 1 < r1) gload(pre, 2 * r0 + 2, r0 + 1, false);  // the next output row's operands travel while this one is multiplied

  const int nt = wave / 3, ky = wave % 3;  // this wave's tiles: dw[32 nt ..][ky][kx = 0, 1, 2][c]
  f32x16 acc[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[kx][r] = 0.0f;

  for (int i = r0; i < r1; ++i) {
    __syncthreads();  // the operands of this output row are visible
    const unsigned char* drow = s_d + (i & 1) * DROW_BYTES;
    const unsigned char* xrow = xslot(2 * i - 1 + ky);
#pragma unroll 2
    for (int s = 0; s < WOUT / 16; ++s) {
      const bf16x8 a = tr_fragment(drow, PXD, nt * 32, s * 16, lane);                 // dy^T: 32 n x 16 positions
      const bf16x8 b0 = tr_fragment(xrow + ODD_OFF, EN, 0, s * 16, lane);             // kx = 0: pixel 2 j - 1 = odd[j]
      const bf16x8 b1 = tr_fragment(xrow, EN, 0, s * 16, lane);                       // kx = 1: pixel 2 j     = even[j]
      const bf16x8 b2 = tr_fragment(xrow + ODD_OFF + EN, EN, 0, s * 16, lane);        // kx = 2: pixel 2 j + 1 = odd[j + 1]
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b2, acc[2], 0, 0, 0);
    }
    __syncthreads();  // every wave has read x rows 2i - 1, 2i and dy row i: their slots take the next operands
    if (i + 1 < r1) {
      lstore(pre, 2 * i + 2, i + 1, true, true);
      if (i + 2 < r1) gload(pre, 2 * i + 4, i + 2, false);
    }
  }
  // C/D map: col = lane & 31 (c), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (n)  ->  slab[workgroup][n][ky][kx][c]
  float* out = slab + (long long)blockIdx.x * DW_ELEMS;
  const int frow = lane & 31, fh = lane >> 5;
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
      out[(n * 9 + ky * 3 + kx) * CI + frow] = acc[kx][r];
    }
}

// 64 gradient elements per workgroup (288 workgroups: every CU has one), the slabs dealt round-robin to its four waves, the four
// partial sums folded through LDS.  (One thread per element over all slabs — 72 workgroups — took 23 us for the 19 MB.)
__global__ __launch_bounds__(256) void conv3x3s2_wgrad_sum_kernel(const float* __restrict__ slab, int n_slabs,
                                                                  unsigned short* __restrict__ dw) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;  // DW_ELEMS is a multiple of 64
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
  int k = wave;
  for (; k + 12 < n_slabs; k += 16) {
    s0 += slab[(long long)k * DW_ELEMS + e];
    s1 += slab[(long long)(k + 4) * DW_ELEMS + e];
    s2 += slab[(long long)(k + 8) * DW_ELEMS + e];
    s3 += slab[(long long)(k + 12) * DW_ELEMS + e];
  }
  for (; k < n_slabs; k += 4) s0 += slab[(long long)k * DW_ELEMS + e];
  part[wave][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wave == 0) dw[e] = __builtin_bit_cast(unsigned short, (__bf16)((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])));
}

inline void plan(int B, int Hout, int* rows, int* per_img) {
  int pi = 256 / B;  // ~256 workgroups (one per CU: 108 KB of LDS), each a whole number of output rows of one image
  if (pi < 1) pi = 1;
  if (pi > Hout) pi = Hout;
  *rows = (Hout + pi - 1) / pi;
  *per_img = (Hout + *rows - 1) / *rows;
}

}  // namespace

extern "C" int hd_conv3x3s2_wgrad_c32_supported(int B, int Hout, int Wout, int Cin, int Cout) {
  return B >= 1 && Hout >= 1 && Hout <= 4096 && Wout == WOUT && Cin == CI && Cout == CO && (long long)B * Hout <= 1 << 20;
}

extern "C" size_t hd_conv3x3s2_wgrad_c32_workspace_bytes(int B, int Hout, int Wout, int Cin, int Cout) {
  if (!hd_conv3x3s2_wgrad_c32_supported(B, Hout, Wout, Cin, Cout)) return 0;
  int rows, per_img;
  plan(B, Hout, &rows, &per_img);
  return (size_t)B * per_img * DW_ELEMS * sizeof(float);
}

extern "C" hd_status hd_conv3x3s2_wgrad_c32(const void* dy, int64_t dy_row_stride, const void* x, int64_t x_row_stride, int B, int Hout,
                                            int Wout, int Cin, int Cout, void* dw, void* workspace, size_t workspace_bytes,
                                            hd_stream stream) {
  if (!dy || !x || !dw || !workspace) return HD_ERR_NULL;
  if (!hd_conv3x3s2_wgrad_c32_supported(B, Hout, Wout, Cin, Cout)) return HD_ERR_SHAPE;
  if (dy_row_stride == 0) dy_row_stride = Cout;
  if (x_row_stride == 0) x_row_stride = Cin;
  if (dy_row_stride < Cout || x_row_stride < Cin || dy_row_stride % 8 || x_row_stride % 8) return HD_ERR_SHAPE;
  if (!hd_aligned16(dy) || !hd_aligned16(x) || !hd_aligned16(dw) || !hd_aligned16(workspace)) return HD_ERR_ALIGN;
  if (workspace_bytes < hd_conv3x3s2_wgrad_c32_workspace_bytes(B, Hout, Wout, Cin, Cout)) return HD_ERR_WORKSPACE;
  int rows, per_img;
  plan(B, Hout, &rows, &per_img);
  const int blocks = B * per_img;
  hipStream_t s = (hipStream_t)stream;
  (void)hd_launch(conv3x3s2_wgrad_c32_kernel, dim3((unsigned)blocks), dim3(NTHREAD), (size_t)LDS_BYTES, s,
                  static_cast<const unsigned short*>(dy), (long long)dy_row_stride, static_cast<const unsigned short*>(x),
                  (long long)x_row_stride, static_cast<float*>(workspace), Hout, rows, per_img);
  HD_RETURN_IF_LAUNCH_FAILED();
  (void)hd_launch(conv3x3s2_wgrad_sum_kernel, dim3((unsigned)(DW_ELEMS / 64)), dim3(256), 0, s,
                  static_cast<const float*>(workspace), blocks, static_cast<unsigned short*>(dw));
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}
