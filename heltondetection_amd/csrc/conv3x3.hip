// conv3x3.hip — 3x3 / pad 1 / stride 1 convolution on channels-last bf16 rows as an implicit GEMM on the matrix cores: the
// forward of the CSP bottlenecks' second convolution and, with flipped + transposed weights, its data gradient
// (hd_conv3x3_fwd).  Round 5 rewrite of the round-4 kernel (one workgroup per CU, 32-pixel tiles per image row: 20 - 26 us per
// layer of the YOLOv5s step at 0.12 of the dense bf16 peak; measured side by side in profiles/r05_conv3.md):
//
//   * FLAT pixel tiles.  The workgroup's ROWS + 2 input rows sit in LDS as ONE image of (W + 2)-pixel rows (one zero pixel
//     of halo on both sides; rows outside the picture stay zero), so an output position q = row (W + 2) + x reads its tap
//     (ky, kx) at the flat pixel q + ky (W + 2) + kx — a constant shift.  M-tiles are 32 CONSECUTIVE flat positions, whatever
//     row they fall in: a 40-pixel row no longer costs two 32-pixel tiles (64 MFMA rows for 40 pixels) but 42 positions, of
//     which the two halo columns are computed and dropped.
//   * 32-channel K chunks (two k-steps of v_mfma_f32_32x32x16_bf16 per tap): a chunk's image is 80 bytes per pixel and its
//     weight slab 64 x 9 x 64 B = 37 KB, so a workgroup needs 55 - 76 KB of LDS and TWO fit a CU: while one stages its next
//     chunk (global -> registers -> LDS, two barriers) the other multiplies.  Cin only has to be a multiple of 32.
//   * every wave takes MT M-tiles x both 32-channel halves of the workgroup's 64 output channels: per (tap, k-step) 2 weight
//     fragments + MT pixel fragments feed 2 MT MFMAs; the fragments of step u + 1 are requested before the MFMAs of step u.
//   * one pass per workgroup (grid = 64-channel tiles x row groups of every image), so the launch is one wave of workgroups.
//
// Epilogue as before: BatchNorm statistics of the bf16-rounded result per lane column, the tile transposed through LDS so that
// the global stores are 16 bytes per lane along whole 128-byte pixel rows.
#include <hip/hip_bf16.h>

#include <cstdlib>

#include "hd_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int kStatReplicas = 16;  // == hd_bn_act_replicas(), checked on the host

__device__ __forceinline__ float bf16_bits_to_float(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

#ifdef HD_CONV3_STAMPS   // diagnostic build only (tools/conv3_stamps.py): phase stamps of wave 0 of every workgroup
__device__ unsigned long long g_conv3_stamps[4096 * 16];
#define HD_STAMP(k)                                                                                         \
  do {                                                                                                      \
    if (tid == 0 && (int)(blockIdx.y * gridDim.x + blockIdx.x) < 4096)                                      \
      g_conv3_stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (k)] = __builtin_amdgcn_s_memrealtime();  \
  } while (0)
#else
#define HD_STAMP(k)
#endif

template <int W, int ROWS, int MT>
struct Conv3Geom {
  static constexpr int WP = W + 2;                 // pixels per LDS row: the image row + one zero pixel on each side
  static constexpr int NSLOT = ROWS + 2;           // input rows held
  static constexpr int KC = 32;                    // input channels per chunk
  static constexpr int PX = KC * 2 + 16;           // bytes per pixel: 80 = 16 x 5, so 16 consecutive pixels at one 16-byte chunk
                                                   // index fall on 16 distinct bank groups (ds_read_b128 is served per 16 lanes)
  static constexpr int MTILES = 4 * MT;            // 32-position tiles of the workgroup (four waves)
  static constexpr int RING_PX = MTILES * 32 + 2 * WP + 2;   // positions a fragment read can touch
  static constexpr int XBYTES = (RING_PX * PX + 15) / 16 * 16;
  static constexpr int WROW = 9 * KC * 2 + 16;     // bytes per output channel of the slab: 592 = 16 x 37
  static constexpr int SLAB = 64 * WROW;
  static constexpr int PATCH_ROW = 64 * 2 + 16;    // epilogue patch: 32 positions x 64 channels
  static constexpr int PATCH = 32 * PATCH_ROW;
  static constexpr int STAT_OFF = XBYTES + (SLAB > 4 * PATCH ? SLAB : 4 * PATCH);   // [2][64] float statistics of the workgroup
  static constexpr int LDS = STAT_OFF + 512;
  static_assert(MTILES * 32 >= ROWS * WP, "the four waves' tiles must cover the pass");
  static_assert(RING_PX >= NSLOT * WP, "the ring must hold every input row");
};

template <int W, int ROWS, int MT, int WPS, bool STATS>   // WPS: workgroups per CU the kernel is built for (registers, LDS)
__global__ __launch_bounds__(256, WPS) void conv3x3_tap_kernel(const unsigned short* __restrict__ X, long long ld_x,
                                                             const unsigned short* __restrict__ Wt,
                                                             unsigned short* __restrict__ Y, long long ld_y, int B, int H,
                                                             int Cin, int Cout, float* __restrict__ stats) {
  using G = Conv3Geom<W, ROWS, MT>;
  constexpr int WP = G::WP, NSLOT = G::NSLOT, KC = G::KC, PX = G::PX, WROW = G::WROW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [pixel ring | weight slab, later the patches]
  unsigned char* xs = smem;
  unsigned char* ws = smem + G::XBYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, fh = lane >> 5;
  const int n0 = blockIdx.x * 64;
  const int ppi = (H + ROWS - 1) / ROWS;  // passes per image
  const int b = (int)blockIdx.y / ppi, r0 = ((int)blockIdx.y % ppi) * ROWS;
  const int NC = Cin / KC;

  // One step = one 32-channel chunk.  The operands of the next TWO chunks travel in registers while the current one is
  // multiplied (the stamps of the diagnostic build showed a chunk's loads taking 1.5 - 2 us to land under load against 0.5 - 1 us
  // of MFMAs per chunk: with one chunk of lookahead every step waited for memory).  Every load is unconditional (clamped row /
  // item index, the value dropped at the LDS store), so the compiler's vmcnt for a set does not drain the younger set.
  constexpr int XCH = NSLOT * W * 4, XIT = (XCH + 255) / 256, WIT = 64 * 9 * 4 / 256;
  constexpr int D = (MT == 1 || (WPS == 1 && MT == 2)) ? 2 : 1;   // chunks of lookahead: two register sets where the register budget holds them
  u32x4 xr[D][XIT], wr[D][WIT];
  const u32x4 zero = {0u, 0u, 0u, 0u};
  auto issue = [&](int c, u32x4 (&xs_)[XIT], u32x4 (&ws_)[WIT]) {
#pragma unroll
    for (int i = 0; i < XIT; ++i) {
      const int id = min(i * 256 + tid, XCH - 1);
      const int slot = id / (W * 4), rem = id % (W * 4), px = rem >> 2, ch = rem & 3;
      const int r = min(max(r0 - 1 + slot, 0), H - 1);
      xs_[i] = *reinterpret_cast<const u32x4*>(X + ((long long)(b * H + r) * W + px) * ld_x + c * KC + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      const int id = i * 256 + tid, co = id / 36, rem = id % 36, tap = rem >> 2, ch = rem & 3;
      ws_[i] = *reinterpret_cast<const u32x4*>(Wt + ((long long)(n0 + co) * 9 + tap) * Cin + c * KC + ch * 8);
    }
  };
  auto stash = [&](const u32x4 (&xs_)[XIT], const u32x4 (&ws_)[WIT]) {
#pragma unroll
    for (int i = 0; i < XIT; ++i) {
      const int id = i * 256 + tid;
      if (XCH % 256 == 0 || id < XCH) {
        const int slot = id / (W * 4), rem = id % (W * 4), px = rem >> 2, ch = rem & 3;
        const int r = r0 - 1 + slot;
        *reinterpret_cast<u32x4*>(xs + (slot * WP + px + 1) * PX + ch * 16) = (r >= 0 && r < H) ? xs_[i] : zero;
      }
    }
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      const int id = i * 256 + tid, co = id / 36, rem = id % 36, tap = rem >> 2, ch = rem & 3;
      *reinterpret_cast<u32x4*>(ws + co * WROW + tap * (KC * 2) + ch * 16) = ws_[i];
    }
  };

  HD_STAMP(0);
  issue(0, xr[0], wr[0]);
  if (D > 1 && NC > 1) issue(1, xr[D - 1], wr[D - 1]);
  for (int i = tid; i < G::XBYTES / 16; i += 256) reinterpret_cast<u32x4*>(xs)[i] = zero;   // halo columns, slack positions
  if (STATS && tid < 128) reinterpret_cast<float*>(smem + G::STAT_OFF)[tid] = 0.0f;

  f32x16 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int mbase = wave * (MT * 32);
  const unsigned char* xa = xs + (mbase + frow) * PX + fh * 16;   // + (32 i + ky WP + kx) PX + 32 sk
  const unsigned char* wa = ws + frow * WROW + fh * 16;           // + 32 j WROW + 64 tap + 32 sk
  auto step = [&](const int c, u32x4 (&xs_)[XIT], u32x4 (&ws_)[WIT]) {
    if (c == 2) HD_STAMP(6);
    __syncthreads();   // the previous chunk has been read (first time: the zero fill is complete)
    if (c == 2) HD_STAMP(7);
    stash(xs_, ws_);
    if (c == 0) HD_STAMP(1);
    if (c == 2) HD_STAMP(8);
    if (c + D < NC) issue(c + D, xs_, ws_);
    __syncthreads();
    if (c == 0) HD_STAMP(2);
    if (c == 2) HD_STAMP(9);
    // 18 steps (tap, k-step); the fragments of step u + PD are requested before the MFMAs of step u.  With one wave per SIMD
    // and 64 cycles of MFMAs per step (MT = 1), one step of lookahead left every step waiting for its LDS reads: the stamps
    // showed the phase at twice the MFMA pace.
    constexpr int PD = MT == 1 ? 3 : 2, NS = PD + 1;
    bf16x8 xf[NS][MT], wf[NS][2];
    auto fetch = [&](const int u) {
      const int slot = u % NS, tap = u >> 1, sk = u & 1, ky = tap / 3, kx = tap % 3;
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[slot][j] = *reinterpret_cast<const bf16x8*>(wa + j * 32 * WROW + tap * (KC * 2) + sk * 32);
#pragma unroll
      for (int i = 0; i < MT; ++i)
        xf[slot][i] = *reinterpret_cast<const bf16x8*>(xa + (i * 32 + ky * WP + kx) * PX + sk * 32);
    };
#pragma unroll
    for (int u = 0; u < PD; ++u) fetch(u);
#pragma unroll
    for (int u = 0; u < 18; ++u) {
      if (u + PD < 18) fetch(u + PD);
      __builtin_amdgcn_sched_barrier(0);   // keep the requests above the multiplies they are meant to overlap
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[u % NS][i], wf[u % NS][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (c == 2) HD_STAMP(10);
  };
  for (int c = 0; c < NC; c += D) {
    step(c, xr[0], wr[0]);
    if (D > 1 && c + 1 < NC) step(c + 1, xr[D - 1], wr[D - 1]);
  }

  HD_STAMP(3);
  // ---- epilogue.  C/D map: column (channel) = lane & 31, row (position) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  // BatchNorm statistics FIRST, straight from the accumulators (of the bf16-rounded values, the ones the normalisation will read):
  // per-lane sums -> LDS -> (behind the barrier the patches need anyway) one global atomic per channel and workgroup, issued
  // BEFORE the tile stores so that the atomics' round trip (the tail of the kernel when they came last: 3 - 4 us per launch
  // against the same kernel without statistics) runs under the stores.
  if (STATS) {
    float s_sum[2] = {0.0f, 0.0f}, s_sq[2] = {0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int q0 = mbase + i * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {   // q0 is wave-uniform: the two half-waves' flags are scalar arithmetic, a lane picks its own
        const int qa = q0 + (r & 3) + 8 * (r >> 2), qb = qa + 4;
        const int rwa = qa / WP, rwb = qb / WP;
        const bool oka = qa - rwa * WP < W && rwa < ROWS && r0 + rwa < H, okb = qb - rwb * WP < W && rwb < ROWS && r0 + rwb < H;
        const bool ok = fh ? okb : oka;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const unsigned short v = __builtin_bit_cast(unsigned short, (__bf16)acc[i][j][r]);
          const float f = ok ? bf16_bits_to_float(v) : 0.0f;
          s_sum[j] += f;
          s_sq[j] = fmaf(f, f, s_sq[j]);
        }
      }
    }
    float* s_st = reinterpret_cast<float*>(smem + G::STAT_OFF);  // [2][64], zeroed with the ring
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float a = s_sum[j] + __shfl_xor(s_sum[j], 32);
      const float q = s_sq[j] + __shfl_xor(s_sq[j], 32);
      if (fh == 0) {
        atomicAdd(&s_st[j * 32 + frow], a);
        atomicAdd(&s_st[64 + j * 32 + frow], q);
      }
    }
  }
  __syncthreads();   // every wave is past its last slab read: the slab area becomes the four patches; the statistics are in LDS
  if (STATS && tid < 128) {
    const float* s_st = reinterpret_cast<const float*>(smem + G::STAT_OFF);
    float* dst = stats + (size_t)((blockIdx.x + blockIdx.y) % kStatReplicas) * 2 * Cout;
    atomicAdd(&dst[(tid / 64) * Cout + n0 + (tid % 64)], s_st[tid]);
  }
  unsigned char* patch = ws + wave * G::PATCH;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int q0 = mbase + i * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int prow = (r & 3) + 8 * (r >> 2) + 4 * fh;
#pragma unroll
      for (int j = 0; j < 2; ++j)
        *reinterpret_cast<unsigned short*>(patch + prow * G::PATCH_ROW + (j * 32 + frow) * 2) =
            __builtin_bit_cast(unsigned short, (__bf16)acc[i][j][r]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // 32 positions x 8 chunks of 16 bytes over 64 lanes
      const int id = k * 64 + lane, prow = id >> 3, ch = id & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(patch + prow * G::PATCH_ROW + ch * 16);
      const int q = q0 + prow, row = q / WP, xo = q - row * WP;
      if (xo < W && row < ROWS && r0 + row < H)
        *reinterpret_cast<uint4*>(Y + ((long long)(b * H + r0 + row) * W + xo) * ld_y + n0 + ch * 8) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  HD_STAMP(4);
  HD_STAMP(5);
}

template <int W, int ROWS, int MT, int WPS>
hd_status launch_conv3x3(const void* x, int64_t ld_x, const void* w, void* y, int64_t ld_y, int B, int H, int Cin, int Cout,
                         float* stats, hipStream_t s) {
  using G = Conv3Geom<W, ROWS, MT>;
  static_assert(G::LDS * WPS <= 160 * 1024, "LDS of the workgroups of one CU");
  const dim3 grid((unsigned)(Cout / 64), (unsigned)(B * ((H + ROWS - 1) / ROWS)));
  const auto* a = static_cast<const unsigned short*>(x);
  const auto* b = static_cast<const unsigned short*>(w);
  auto* c = static_cast<unsigned short*>(y);
  if (stats)
    (void)hd_launch(conv3x3_tap_kernel<W, ROWS, MT, WPS, true>, grid, dim3(256), (size_t)G::LDS, s, a, (long long)ld_x, b, c, (long long)ld_y, B, H,
                    Cin, Cout, stats);
  else
    (void)hd_launch(conv3x3_tap_kernel<W, ROWS, MT, WPS, false>, grid, dim3(256), (size_t)G::LDS, s, a, (long long)ld_x, b, c, (long long)ld_y, B,
                    H, Cin, Cout, (float*)nullptr);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

// Row widths with an instantiation: the 640^2 / 1280^2 pyramids (20, 40, 80) and the 832^2 ones (26, 52, 104).
inline bool conv3x3_width_ok(int W) { return W == 20 || W == 26 || W == 40 || W == 52 || W == 80 || W == 104; }

}  // namespace

#ifdef HD_CONV3_STAMPS
extern "C" int hd_conv3_stamps_read(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_conv3_stamps), sizeof(unsigned long long) * n);
}
#endif

extern "C" int hd_bn_act_replicas(void);

// conv3x3_c32.hip: 32 -> 32 channels on 160-pixel rows (the first C3 block's bottleneck), a row-walk kernel of its own
int hd_conv3x3_c32_supported(int B, int H, int W, int Cin, int Cout);
hd_status hd_conv3x3_c32(const void* x, int64_t ldx, const void* w, int B, int H, void* y, int64_t ldy, float* stats, hipStream_t s);

extern "C" int hd_conv3x3_fwd_supported(int B, int H, int W, int Cin, int Cout) {
  if (hd_conv3x3_c32_supported(B, H, W, Cin, Cout)) return 1;
  return B >= 1 && H >= 1 && conv3x3_width_ok(W) && Cin >= 32 && Cin % 32 == 0 && Cout >= 64 && Cout % 64 == 0 &&
         Cin <= 1024 && Cout <= 1024 && (long long)B * H * W <= 0x7fffffff && (long long)B * H <= 65535;
}

extern "C" hd_status hd_conv3x3_fwd(const void* x, int64_t x_row_stride, const void* w, int B, int H, int W, int Cin, int Cout,
                                    void* y, int64_t y_row_stride, float* stats, hd_stream stream) {
  if (!x || !w || !y) return HD_ERR_NULL;
  if (!hd_conv3x3_fwd_supported(B, H, W, Cin, Cout)) return HD_ERR_SHAPE;
  if (x_row_stride == 0) x_row_stride = Cin;
  if (y_row_stride == 0) y_row_stride = Cout;
  if (x_row_stride < Cin || y_row_stride < Cout || x_row_stride % 8 || y_row_stride % 8) return HD_ERR_SHAPE;
  if (!hd_aligned16(x) || !hd_aligned16(w) || !hd_aligned16(y)) return HD_ERR_ALIGN;
  if (stats && hd_bn_act_replicas() != kStatReplicas) return HD_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (hd_conv3x3_c32_supported(B, H, W, Cin, Cout)) return hd_conv3x3_c32(x, x_row_stride, w, B, H, y, y_row_stride, stats, s);
  // <W, ROWS, MT, workgroups per CU>: ROWS x (W + 2) flat positions per workgroup in 4 MT tiles of 32.  Measured per shape
  // (tools/conv3_probe.py, profiles/r05_conv3.md): 80 / 104-pixel rows two workgroups per CU with MT = 2 (MT = 4 alone on the CU:
  // slower), 52-pixel rows likewise (25.4 / 21.0 us against 31.6 / 26.3 with MT = 1), 40-pixel rows one workgroup of MT = 2 per CU
  // (224 workgroups; two of MT = 1: equal to 4 % slower), 20 / 26-pixel rows MT = 1 (MT = 2 would leave half the CUs idle).
  switch (W) {
    case 104: return launch_conv3x3<104, 2, 2, 2>(x, x_row_stride, w, y, y_row_stride, B, H, Cin, Cout, stats, s);
    case 80: return launch_conv3x3<80, 3, 2, 2>(x, x_row_stride, w, y, y_row_stride, B, H, Cin, Cout, stats, s);
    case 52: return launch_conv3x3<52, 4, 2, 2>(x, x_row_stride, w, y, y_row_stride, B, H, Cin, Cout, stats, s);
    case 40: return launch_conv3x3<40, 6, 2, 1>(x, x_row_stride, w, y, y_row_stride, B, H, Cin, Cout, stats, s);
    case 26: return launch_conv3x3<26, 4, 1, 2>(x, x_row_stride, w, y, y_row_stride, B, H, Cin, Cout, stats, s);
    case 20: return launch_conv3x3<20, 5, 1, 2>(x, x_row_stride, w, y, y_row_stride, B, H, Cin, Cout, stats, s);
    default: return HD_ERR_SHAPE;
  }
}
