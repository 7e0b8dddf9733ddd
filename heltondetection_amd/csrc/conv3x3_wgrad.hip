// conv3x3_wgrad.hip — weight gradient of the 3x3 / pad 1 convolutions, stride 1 and 2, on channels-last bf16 rows:
//     dW[n][ky][kx][c] = sum over (b, y, x) of dY[b, y, x, n] * X[b, S y + ky - 1, S x + kx - 1, c]
// Round 5 rewrite of conv3x3_wgrad_kernel / conv3x3s2_wgrad_kernel (pointwise_conv.hip: one image row per barrier pair, 64 x 64
// (n, c) tiles, a 147 KB float32 partial per workgroup — 23 - 50 us per layer at 0.11 of the dense bf16 peak, bound by the row
// barriers and by writing and re-reading 19 - 38 MB of partials per layer).  The same flat-position idea as csrc/conv3x3.hip:
//
//   * a BLOCK of R output rows of one image sits in LDS as flat images of 32-channel positions (64-byte rows: four consecutive
//     rows are the four quarters of a bank row, so ds_read_b64_tr_b16 needs no padding): dY as [R][WP] positions with ZERO halo
//     columns, X as [R + 2][WP] with one zero pixel on both sides (stride 1, WP = W + 2) or as four parity planes
//     [row parity][column parity][R + 1][WP] (stride 2, WP = W + 1: even / odd input rows and columns de-interleaved), so that
//     EVERY tap is the dY image against the X image shifted by a constant number of positions.  The reduction runs over the
//     flat positions in steps of 16, across row ends; one barrier pair per block instead of one per image row.
//   * 32 x 32 (n, c) tiles for all nine taps: a partial is 37 KB instead of 147 KB (the slab of a 128 -> 128 layer at 512
//     workgroups: 18.9 MB -> 9.4 MB at 256 partials), channel counts only have to be multiples of 32 (the 32-channel layers
//     at 160 x 160 come off the library), and W is a run-time value (any width whose block fits LDS, 208-pixel rows included).
//   * four waves = two tap groups (taps 0-4 / 5-8) x two halves of the position steps; the halves are folded through LDS.
//   * no register staging: a block is copied global -> LDS by all threads, two or three workgroups per CU overlap each other's
//     copy and multiply phases.
// The slice sum of the PREVIOUS weight-gradient launch rides in extra workgroups, as in every weight-gradient kernel here.
#include <hip/hip_bf16.h>

#include <cstdlib>

#include "hd_bn_fuse.h"
#include "hd_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int kPos = 64;   // bytes per position: 32 channels

#ifdef HD_CONV3_STAMPS   // diagnostic build only (tools/wgrad3_stamps.py): phase stamps of thread 0 of every workgroup
__device__ unsigned long long g_wgrad3_stamps[4096 * 16];
#define HD_STAMP(k)                                                                                          \
  do {                                                                                                       \
    if (threadIdx.x == 0 && (int)(blockIdx.y * gridDim.x + blockIdx.x) < 4096)                               \
      g_wgrad3_stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = __builtin_amdgcn_s_memrealtime();  \
  } while (0)
#else
#define HD_STAMP(k)
#endif

// transpose fragment of a [position][32 channels] image: 16 positions from m_base, all 32 channels — lane l gets channel
// l & 31, positions m_base + 8 (l >> 5) + j
__device__ __forceinline__ int tr_lane_offset(int lane) {   // the lane's part of a fragment address
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  return (8 * (g >> 1) + q) * kPos + (16 * (g & 1) + 4 * p) * 2;
}
__device__ __forceinline__ bf16x8 tr_fragment(const unsigned char* img, int m_base, int lane_off) {
  const unsigned char* a = img + m_base * kPos + lane_off;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * kPos));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

struct FlatGeom {   // positions, in units of kPos bytes
  int WP;           // positions per image row
  int dy_pos;       // dY image (a multiple of 16)
  int plane;        // one X plane (stride 1: the only one)
  int x_pos;        // all X planes
};
__host__ __device__ inline FlatGeom flat_geom(int S, int Wo, int R) {
  FlatGeom g;
  g.WP = S == 1 ? Wo + 2 : Wo + 1;
  g.dy_pos = (R * g.WP + 15) / 16 * 16;
  g.plane = S == 1 ? g.dy_pos + 2 * g.WP + 2 : g.dy_pos + g.WP + 1;
  if (g.plane < (S == 1 ? (R + 2) * g.WP : (R + 1) * g.WP)) g.plane = S == 1 ? (R + 2) * g.WP : (R + 1) * g.WP;
  g.plane = (g.plane + 3) / 4 * 4;                         // 256-byte aligned planes
  g.x_pos = S == 1 ? g.plane : 4 * g.plane;
  return g;
}

template <int S>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_flat_kernel(const unsigned short* __restrict__ dY, long long ld_dy,
                                                                    const unsigned short* __restrict__ X, long long ld_x, int B, int Ho,
                                                                    int Wo, int N, int K, int R, int rows_per_slice,
                                                                    float* __restrict__ slab, int n_slices, const hdfuse::ReduceJob job) {
  __shared__ float4 s_part[16][17];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [dY image | X image / planes], later the fold area
  if ((int)blockIdx.y >= n_slices) {  // the slice sum of the previous weight-gradient launch rides along (hd_wgrad_job)
    const long long blk = (long long)(blockIdx.y - n_slices) * gridDim.x + blockIdx.x;
    if (blk < hdfuse::reduce_blocks(job.count)) hdfuse::reduce_block(job, blk, s_part);
    return;
  }
  const FlatGeom g = flat_geom(S, Wo, R);
  const int WP = g.WP;
  unsigned char* ys = smem;
  unsigned char* xs = smem + (size_t)g.dy_pos * kPos;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wave & 1, ph = wave >> 1;          // tap group (taps 0-4 / 5-8), half of the position steps
  const int tiles_k = K / 32;
  const int n0 = ((int)blockIdx.x / tiles_k) * 32, k0 = ((int)blockIdx.x % tiles_k) * 32;
  const int Hi = S * Ho, Wi = S * Wo;
  const long long rows_total = (long long)B * Ho;
  const long long first = (long long)blockIdx.y * rows_per_slice;
  const long long last = first + rows_per_slice < rows_total ? first + rows_per_slice : rows_total;

  HD_STAMP(0);
  const u32x4 zero = {0u, 0u, 0u, 0u};
  for (int i = tid; i < (g.dy_pos + g.x_pos) * (kPos / 16); i += 256) reinterpret_cast<u32x4*>(smem)[i] = zero;   // halos, slack

  // per-tap view of the X image: plane base + constant shift (in positions)
  int tap_off[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const int tap = tg * 5 + t < 9 ? tg * 5 + t : 8, ky = tap / 3, kx = tap % 3;
    if (S == 1) tap_off[t] = ky * WP + kx;
    else tap_off[t] = ((ky != 1) * 2 + (kx != 1)) * g.plane + (ky >= 1) * WP + (kx >= 1);
  }
  const int ntap = tg == 0 ? 5 : 4;
  const int lane_off = tr_lane_offset(lane);

  f32x16 acc[5];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  for (long long row = first; row < last;) {
    const int b = (int)(row / Ho), y0 = (int)(row - (long long)b * Ho);
    int nr = R < Ho - y0 ? R : Ho - y0;
    if ((long long)nr > last - row) nr = (int)(last - row);
    __syncthreads();   // the previous block has been read (first time: the zero fill is complete)
    if (row == first) HD_STAMP(1);
    // ---- the block's operands, global -> registers -> LDS.  Up to kInFlight 16-byte items per thread are requested before the
    // first of them is stored (a round of 4 paid a full memory latency per 4 KB of a 50 - 90 KB block: 4 - 8 us per block
    // against 1 us of MFMAs); dY first (rows past nr: zeros, a shorter block must not see the previous block's rows), then X
    // (rows outside the picture: zeros).  Item -> (row, pixel, chunk) is computed once and kept for the store.
    {
      constexpr int kInFlight = 12;
      const int y_items = R * Wo * 4;
      const int xrows = S == 1 ? nr + 2 : 2 * nr + 1;
      const int x_items = xrows * Wi * 4;
      const int items = y_items + x_items;
      const int i_first = S * y0 - 1;
      const unsigned short* ybase = dY + ((long long)(b * Ho + y0) * Wo) * ld_dy + n0;
      for (int i0 = tid; i0 < items; i0 += 256 * kInFlight) {
        u32x4 v[kInFlight];
        int dst[kInFlight];
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
          const int id = i0 + u * 256;
          v[u] = zero;
          dst[u] = -1;
          if (id < y_items) {
            const int r = id / (Wo * 4), rem = id - r * (Wo * 4);
            dst[u] = (r * WP + (rem >> 2)) * kPos + (rem & 3) * 16;
            if (r < nr) v[u] = *reinterpret_cast<const u32x4*>(ybase + ((long long)r * Wo + (rem >> 2)) * ld_dy + (rem & 3) * 8);
          } else if (id < items) {
            const int ix = id - y_items;
            const int ri = ix / (Wi * 4), rem = ix - ri * (Wi * 4), xi = rem >> 2, ii = i_first + ri;
            int pos;
            if (S == 1) pos = ri * WP + xi + 1;
            else pos = ((ii & 1) * 2 + (xi & 1)) * g.plane + ((ii >> 1) - (y0 - 1)) * WP + (xi >> 1) + 1;
            dst[u] = (g.dy_pos + pos) * kPos + (rem & 3) * 16;
            if (ii >= 0 && ii < Hi)
              v[u] = *reinterpret_cast<const u32x4*>(X + ((long long)(b * Hi + ii) * Wi + xi) * ld_x + k0 + (rem & 3) * 8);
          }
        }
#pragma unroll
        for (int u = 0; u < kInFlight; ++u)
          if (dst[u] >= 0) *reinterpret_cast<u32x4*>(smem + dst[u]) = v[u];
      }
    }
    if (row == first) HD_STAMP(2);
    __syncthreads();
    if (row == first) HD_STAMP(3);
    // ---- multiply: steps of 16 positions, this wave's half of them, its taps
    const int steps = (nr * WP + 15) / 16;
    bf16x8 a = tr_fragment(ys, ph * 16, lane_off), bq[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) bq[t] = tr_fragment(xs, ph * 16 + tap_off[t], lane_off);
    for (int s = ph; s < steps; s += 2) {
      const int sn = s + 2 < steps ? s + 2 : s;   // the next step's fragments are requested before this step's MFMAs
      const bf16x8 an = tr_fragment(ys, sn * 16, lane_off);
      bf16x8 bn[5];
#pragma unroll
      for (int t = 0; t < 5; ++t) bn[t] = tr_fragment(xs, sn * 16 + tap_off[t], lane_off);
#pragma unroll
      for (int t = 0; t < 5; ++t)
        if (t < 4 || ntap == 5) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq[t], acc[t], 0, 0, 0);
      a = an;
#pragma unroll
      for (int t = 0; t < 5; ++t) bq[t] = bn[t];
    }
    if (row == first) HD_STAMP(4);
    row += nr;
  }
  HD_STAMP(5);

  // ---- fold the two position halves through LDS, then one partial per workgroup: slab[slice][n][tap][c]
  __syncthreads();
  float* fold = reinterpret_cast<float*>(smem) + (size_t)tg * 5 * 16 * 64;
  if (ph == 1) {
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) fold[(t * 16 + r) * 64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (ph == 1) return;
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] += fold[(t * 16 + r) * 64 + lane];
  // C/D map: column (c) = lane & 31, row (n) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float* out = slab + (size_t)blockIdx.y * N * 9 * K;
  const int c = k0 + (lane & 31);
  const int nb = n0 + 4 * (lane >> 5);
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    if (t < 4 || ntap == 5) {
      const int tap = tg * 5 + t;
#pragma unroll
      for (int r = 0; r < 16; ++r) out[((size_t)(nb + (r & 3) + 8 * (r >> 2)) * 9 + tap) * K + c] = acc[t][r];
    }
  }
  HD_STAMP(6);
}

struct FlatPlan {
  int tiles, R, rows_per_slice, slices;
  size_t lds;
};

inline size_t flat_lds(int S, int Wo, int R) {
  const FlatGeom g = flat_geom(S, Wo, R);
  const size_t img = (size_t)(g.dy_pos + g.x_pos) * kPos;
  const size_t fold = 2 * 5 * 16 * 64 * sizeof(float);   // the two tap groups' upper halves
  return img > fold ? img : fold;
}

// Row block R: the largest whose images leave room for two workgroups per CU (<= 76 KB), at least 1 (a wide row alone may
// need more: then one workgroup per CU).  Slices: about `want` workgroups in all (two per CU), each a whole number of
// blocks where that is possible.
inline FlatPlan flat_plan(int B, int Ho, int Wo, int Cin, int Cout, int S) {
  FlatPlan p{};
  p.tiles = (Cout / 32) * (Cin / 32);
  const int want = 512;   // two workgroups per CU (256 measured within 5 % either way, profiles/r05_wgrad3.md)
  const long long rows_total = (long long)B * Ho;
  int rmax = 1;
  while (rmax < Ho && rmax < 64 && flat_lds(S, Wo, rmax + 1) <= 76 * 1024) ++rmax;
  long long slices = want / p.tiles;
  if (slices < 1) slices = 1;
  if (slices > rows_total) slices = rows_total;
  long long rps = (rows_total + slices - 1) / slices;
  const int blocks = (int)((rps + rmax - 1) / rmax);       // blocks per slice
  p.R = (int)((rps + blocks - 1) / blocks);                // equal blocks instead of rmax, rmax, ..., remainder
  p.rows_per_slice = (int)rps;
  p.slices = (int)((rows_total + rps - 1) / rps);
  p.lds = flat_lds(S, Wo, p.R);
  return p;
}

unsigned piggyback_rows(const hdfuse::ReduceJob& job, unsigned grid_x) {
  if (!job.slab) return 0;
  const long long blocks = hdfuse::reduce_blocks(job.count);
  return (unsigned)((blocks + grid_x - 1) / grid_x);
}

}  // namespace

#ifdef HD_CONV3_STAMPS
extern "C" int hd_wgrad3_stamps_read(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wgrad3_stamps), sizeof(unsigned long long) * n);
}
#endif

// the row kernels (pointwise_conv.hip)
int hd_conv3x3_wgrad_supported_v1(int B, int H, int W, int Cin, int Cout, int stride);
size_t hd_conv3x3_wgrad_workspace_bytes_v1(int B, int H, int W, int Cin, int Cout, int stride);
hd_status hd_conv3x3_wgrad_partial_v1(const void* dy, int64_t dy_row_stride, const void* x, int64_t x_row_stride, int B, int H, int W,
                                      int Cin, int Cout, int stride, void* dw, void* workspace, size_t workspace_bytes,
                                      const hd_wgrad_job* piggyback, hd_wgrad_job* job_out, hd_stream stream);
// Which kernel takes a shape.  Measured per YOLOv5s layer (tools/wgrad3_probe.py, profiles/r05_wgrad3.md): the 64 x 64-tile row
// kernels of round 4 read every operand in whole 128-byte lines, once per 64-channel tile; the flat kernel's 32 x 32 tiles read
// 64-byte half lines once per 32-channel tile — 4 x the cache-line traffic, which overflows the 4 MB L2 of an XCD on the 40- and
// 80-pixel levels (stamps: 6.7 - 8 us to copy a 56 KB block).  The flat kernel had won where the picture is small and the
// channel count large (20 x 20, 256 -> 256, stride 1: 36 - 40 us against the pair-form row kernel's 52); without the pair form
// the row kernel runs that shape in 32 us, so the flat kernel is now the one for what the row kernels do not take: 32-channel
// layers and widths other than 20 / 26 / 40 / 52 / 80 / 104.  hd_conv3x3_wgrad_select forces one or the other (tests, probes).
static int g_wgrad3_select = 0;   // hd_conv3x3_wgrad_select: 0 auto, 1 the row kernels only, 2 the flat kernel wherever it can run
static bool wgrad3_use_v1(int B, int H, int W, int Cin, int Cout, int stride) {
  if (g_wgrad3_select == 1) return true;
  if (g_wgrad3_select == 2) return false;
  if (!hd_conv3x3_wgrad_supported_v1(B, H, W, Cin, Cout, stride)) return false;
  // every shape the row kernels take: since their 512-thread pair form went (round 5) the one case the flat kernel had won —
  // 16 x 256 x 20 x 20 -> 256, stride 1: 36 - 40 us flat against 52 — runs 32 us on them (profiles/r05b_wgrad_rows.md)
  return true;
}
#define wgrad3_v1() wgrad3_use_v1(B, H, W, Cin, Cout, stride)

extern "C" void hd_conv3x3_wgrad_select(int mode) { g_wgrad3_select = mode >= 0 && mode <= 2 ? mode : 0; }

extern "C" int hd_conv3x3_wgrad_supported(int B, int H, int W, int Cin, int Cout, int stride) {
  if (wgrad3_v1()) return hd_conv3x3_wgrad_supported_v1(B, H, W, Cin, Cout, stride);
  if (!((stride == 1 || stride == 2) && B >= 1 && H >= 1 && W >= 4 && Cin >= 32 && Cin % 32 == 0 && Cout >= 32 && Cout % 32 == 0 &&
        Cin <= 1024 && Cout <= 1024 && (long long)B * H * W * stride * stride <= 0x7fffffff))
    return 0;
  if (!(flat_lds(stride, W, 1) <= 150 * 1024 && (long long)(Cin / 32) * (Cout / 32) <= 65535)) return 0;
  if (g_wgrad3_select == 2) return 1;   // tests / probes: every shape the kernel can run
  // Only where it is measured to beat what it would replace (profiles/r05_wgrad3.md).  Shapes the row kernels do not take: narrow
  // layers (<= 128 channels: the 32 -> 32 layer at 160 x 160 runs 28 - 30 us against the library's 50 + zero-fill + cast) on
  // small pictures; NOT the 256 / 512-channel layers of config 3 (32 x 32 tiles make 64 - 256 tiles there, each walking a long
  // chain of blocks: the FasterRCNN step lost 2.9 ms when they were taken) and not wide stride-2 layers (4 x the input of
  // their output).
  const bool narrow = Cin <= 128 && Cout <= 128;
  if (hd_conv3x3_wgrad_supported_v1(B, H, W, Cin, Cout, stride)) return 1;   // the one 256 -> 256 case of wgrad3_use_v1
  return stride == 1 ? ((W <= 52 && narrow) || (Cin == 32 && Cout == 32)) : (W <= 26 && narrow);
}

extern "C" size_t hd_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int stride) {
  if (wgrad3_v1()) return hd_conv3x3_wgrad_workspace_bytes_v1(B, H, W, Cin, Cout, stride);
  if (!hd_conv3x3_wgrad_supported(B, H, W, Cin, Cout, stride)) return 0;
  return (size_t)flat_plan(B, H, W, Cin, Cout, stride).slices * Cout * 9 * Cin * sizeof(float);
}

extern "C" hd_status hd_conv3x3_wgrad_partial(const void* dy, int64_t dy_row_stride, const void* x, int64_t x_row_stride, int B,
                                              int H, int W, int Cin, int Cout, int stride, void* dw, void* workspace,
                                              size_t workspace_bytes, const hd_wgrad_job* piggyback, hd_wgrad_job* job_out,
                                              hd_stream stream) {
  if (wgrad3_v1())
    return hd_conv3x3_wgrad_partial_v1(dy, dy_row_stride, x, x_row_stride, B, H, W, Cin, Cout, stride, dw, workspace, workspace_bytes,
                                       piggyback, job_out, stream);
  if (!dy || !x || !dw || !workspace || !job_out) return HD_ERR_NULL;
  if (!hd_conv3x3_wgrad_supported(B, H, W, Cin, Cout, stride)) return HD_ERR_SHAPE;
  if (dy_row_stride == 0) dy_row_stride = Cout;
  if (x_row_stride == 0) x_row_stride = Cin;
  if (dy_row_stride < Cout || x_row_stride < Cin || dy_row_stride % 8 || x_row_stride % 8) return HD_ERR_SHAPE;
  if (!hd_aligned16(dy) || !hd_aligned16(x) || !hd_aligned16(dw) || !hd_aligned16(workspace)) return HD_ERR_ALIGN;
  const FlatPlan p = flat_plan(B, H, W, Cin, Cout, stride);
  if (workspace_bytes < (size_t)p.slices * Cout * 9 * Cin * sizeof(float)) return HD_ERR_WORKSPACE;
  hdfuse::ReduceJob jv{};
  if (piggyback) {
    if (!piggyback->slab || !piggyback->dw) return HD_ERR_NULL;
    if (piggyback->slices < 1 || piggyback->count < 4 || piggyback->count % 4) return HD_ERR_SHAPE;
    if (!hd_aligned16(piggyback->slab) || (reinterpret_cast<uintptr_t>(piggyback->dw) & 7u)) return HD_ERR_ALIGN;
    jv.slab = piggyback->slab;
    jv.slices = piggyback->slices;
    jv.count = piggyback->count;
    jv.dw = static_cast<unsigned short*>(piggyback->dw);
  }
  hipStream_t s = (hipStream_t)stream;
  const auto* a = static_cast<const unsigned short*>(dy);
  const auto* b = static_cast<const unsigned short*>(x);
  float* slab = static_cast<float*>(workspace);
  const dim3 grid((unsigned)p.tiles, (unsigned)p.slices + piggyback_rows(jv, (unsigned)p.tiles));
  if (stride == 1)
    (void)hd_launch(conv3x3_wgrad_flat_kernel<1>, grid, dim3(256), p.lds, s, a, (long long)dy_row_stride, b, (long long)x_row_stride, B, H, W,
                    Cout, Cin, p.R, p.rows_per_slice, slab, p.slices, jv);
  else
    (void)hd_launch(conv3x3_wgrad_flat_kernel<2>, grid, dim3(256), p.lds, s, a, (long long)dy_row_stride, b, (long long)x_row_stride, B, H, W,
                    Cout, Cin, p.R, p.rows_per_slice, slab, p.slices, jv);
  HD_RETURN_IF_LAUNCH_FAILED();
  job_out->slab = slab;
  job_out->slices = p.slices;
  job_out->count = (int64_t)Cout * 9 * Cin;
  job_out->dw = dw;
  return HD_OK;
}

extern "C" hd_status hd_wgrad_reduce(const hd_wgrad_job* job, hd_stream stream);

extern "C" hd_status hd_conv3x3_wgrad(const void* dy, int64_t dy_row_stride, const void* x, int64_t x_row_stride, int B, int H,
                                      int W, int Cin, int Cout, int stride, void* dw, void* workspace, size_t workspace_bytes,
                                      hd_stream stream) {
  hd_wgrad_job job;
  const hd_status st = hd_conv3x3_wgrad_partial(dy, dy_row_stride, x, x_row_stride, B, H, W, Cin, Cout, stride, dw, workspace,
                                                workspace_bytes, nullptr, &job, stream);
  return st != HD_OK ? st : hd_wgrad_reduce(&job, stream);
}
