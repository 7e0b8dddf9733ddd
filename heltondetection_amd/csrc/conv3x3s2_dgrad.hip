// conv3x3s2_dgrad.hip — data gradient of the FIRST stride-2 3x3 convolution of the YOLOv5 backbone (32 <- 64 channels, 320^2 <- 160^2
// at 640^2): hd_conv3x3s2_dgrad.
//
// The layer is memory-bound by a wide margin (15 GFLOP against 52 MB of dy in and 105 MB of dx out: 25 us at the copy rate, 6 us of
// MFMA), and the library spends 76 us on it plus a 25 us zero-fill of dx (its kernel accumulates into dx).  Here every dx
// element is written exactly once, by the workgroup that owns its dy row:
//     dx[2i  ][2j  ] = dy[i][j] W11
//     dx[2i  ][2j+1] = dy[i][j] W12 + dy[i][j+1] W10
//     dx[2i+1][2j  ] = dy[i][j] W21 + dy[i+1][j] W01
//     dx[2i+1][2j+1] = dy[i][j] W22 + dy[i][j+1] W20 + dy[i+1][j] W02 + dy[i+1][j+1] W00        (Wkykx: [co] -> [ci])
// i.e. the four output parities are 1 / 2 / 2 / 4 taps over the 2 x 2 dy neighbourhood: 9 (tap, 64 x 32) products per dy
// position, no multiplication by an inserted zero.  dy rows / columns past the edge read as zeros (halo pixel, zero row).
//   * a workgroup (5 waves = the 5 x 32 positions of a dy row) walks down `rows_per_wg` dy rows of one image with a ring of
//     three rows in LDS (pixel stride 144 bytes: the A fragments of 16 consecutive pixels fall on 16 bank groups) and the next
//     row already travelling in registers: one barrier per row, every dy row read from memory once (+ one halo row per walk);
//   * the nine tap matrices sit in LDS transposed to [tap][ci][co] (41 KB, built once per workgroup from the layer's own
//     [co][tap][ci] weight: no flipped copy is needed);
//   * the two dx rows of a dy row leave through a wave-private patch as 16-byte stores: 4 KB contiguous per wave and row.
#include "hd_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int CO = 64, CI = 32, WOUT = 160;
constexpr int NWAVE = WOUT / 32, NTHREAD = NWAVE * 64;
constexpr int PX = CO * 2 + 16;                  // bytes per dy pixel in LDS
constexpr int ROW_BYTES = (WOUT + 1) * PX;       // + the zero pixel right of the row
constexpr int W_ROW = CO * 2 + 16;               // bytes per (tap, ci) row of co
constexpr int W_BYTES = 9 * CI * W_ROW;
constexpr int PATCH_ROW = 2 * CI * 2 + 16;       // one dy position = two dx pixels
constexpr int PATCH_BYTES = 32 * PATCH_ROW;
constexpr int LDS_BYTES = W_BYTES + 3 * ROW_BYTES + NWAVE * PATCH_BYTES;
constexpr int CHUNKS = WOUT * CO / 8;            // 16-byte chunks per dy row
constexpr int LD_IT = CHUNKS / NTHREAD;
static_assert(CHUNKS % NTHREAD == 0, "row chunks per thread");

__global__ __launch_bounds__(NTHREAD) void conv3x3s2_dgrad_c32_kernel(const unsigned short* __restrict__ dy, long long ld_dy,
                                                                       const unsigned short* __restrict__ w,
                                                                       unsigned short* __restrict__ dx, long long ld_dx, int H,
                                                                       int rows_per_wg, int per_img) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [tap matrices | ring of 3 dy rows | wave patches]
  unsigned char* s_w = smem;
  unsigned char* s_ring = smem + W_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 31, fh = lane >> 5;
  unsigned char* patch = s_ring + 3 * ROW_BYTES + wave * PATCH_BYTES;
  const int b = blockIdx.x / per_img;
  const int r0 = (blockIdx.x % per_img) * rows_per_wg;
  const int r1 = r0 + rows_per_wg < H ? r0 + rows_per_wg : H;  // rows [r0, r1) are this workgroup's; row r1 is its halo
  const unsigned short* dyb = dy + (long long)b * H * WOUT * ld_dy;
  unsigned short* dxb = dx + (long long)b * (2 * H) * (2 * WOUT) * ld_dx;

  u32x4 pre[LD_IT];
  auto gload = [&](u32x4 (&pre)[LD_IT], int row) {  // rows >= H are zeros (uniform branch)
    if (row < H) {
#pragma unroll
      for (int p = 0; p < LD_IT; ++p) {
        const int id = p * NTHREAD + tid;
        pre[p] = *reinterpret_cast<const u32x4*>(dyb + ((long long)row * WOUT + (id >> 3)) * ld_dy + (id & 7) * 8);
      }
    } else {
#pragma unroll
      for (int p = 0; p < LD_IT; ++p) pre[p] = u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto lstore = [&](const u32x4 (&pre)[LD_IT], int row) {
    unsigned char* dst = s_ring + (row % 3) * ROW_BYTES;
#pragma unroll
    for (int p = 0; p < LD_IT; ++p) {
      const int id = p * NTHREAD + tid;
      *reinterpret_cast<u32x4*>(dst + (id >> 3) * PX + (id & 7) * 16) = pre[p];
    }
  };

  // Prologue: the weight's chunks (8 per thread) and dy rows r0, r0 + 1 are all requested before the first LDS store (as a
  // load -> store loop the weight alone was eight L2 round trips per workgroup: see conv3x3s2_fwd.hip).
  constexpr int WCH = CO * 9 * (CI / 8), WIT = (WCH + NTHREAD - 1) / NTHREAD;
  u32x4 wreg[WIT], pre_b[LD_IT];
#pragma unroll
  for (int p = 0; p < WIT; ++p) {
    const int id = p * NTHREAD + tid < WCH ? p * NTHREAD + tid : WCH - 1;  // threads past the end re-read the last chunk (never stored)
    const int co = id / (9 * (CI / 8)), rem = id % (9 * (CI / 8)), tap = rem / (CI / 8), c8 = rem % (CI / 8);
    wreg[p] = *reinterpret_cast<const u32x4*>(w + ((long long)co * 9 + tap) * CI + c8 * 8);
  }
  gload(pre, r0);
  gload(pre_b, r0 + 1);
  if (tid < 3 * 8)  // the zero pixel right of each ring row (never overwritten)
    *reinterpret_cast<u32x4*>(s_ring + (tid >> 3) * ROW_BYTES + WOUT * PX + (tid & 7) * 16) = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int p = 0; p < WIT; ++p) {  // w[co][tap][ci] -> s_w[tap][ci][co]
    const int id = p * NTHREAD + tid;
    if (id < WCH) {
      const int co = id / (9 * (CI / 8)), rem = id % (9 * (CI / 8)), tap = rem / (CI / 8), c8 = rem % (CI / 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned short x = (unsigned short)(wreg[p][e >> 1] >> ((e & 1) * 16));
        *reinterpret_cast<unsigned short*>(s_w + (tap * CI + c8 * 8 + e) * W_ROW + co * 2) = x;
      }
    }
  }
  lstore(pre, r0);
  lstore(pre_b, r0 + 1);
  if (r0 + 2 <= r1) gload(pre, r0 + 2);

  const int j0 = wave * 32;
  for (int i = r0; i < r1; ++i) {
    __syncthreads();  // the previous row's reads are done (its slot is free), the rows stored since are visible
    if (i + 2 <= r1) lstore(pre, i + 2);
    if (i + 3 <= r1) gload(pre, i + 3);
    const unsigned char* cur = s_ring + (i % 3) * ROW_BYTES + (j0 + frow) * PX + fh * 16;
    const unsigned char* nxt = s_ring + ((i + 1) % 3) * ROW_BYTES + (j0 + frow) * PX + fh * 16;
    const unsigned char* wl = s_w + frow * W_ROW + fh * 16;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.0f;
    // The 4 + 9 fragments of k-step s + 1 are requested before the nine MFMAs of k-step s issue (left to the compiler: read, wait,
    // multiply per fragment — an LDS round trip per MFMA, which was where the first version spent its time, not in memory).
    struct Frags {
      bf16x8 a00, a01, a10, a11, t[9];
    };
    Frags fr[2];
    auto fetch = [&](const int s, Frags& f) {
      f.a00 = *reinterpret_cast<const bf16x8*>(cur + s * 32);
      f.a01 = *reinterpret_cast<const bf16x8*>(cur + PX + s * 32);
      f.a10 = *reinterpret_cast<const bf16x8*>(nxt + s * 32);
      f.a11 = *reinterpret_cast<const bf16x8*>(nxt + PX + s * 32);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) f.t[tap] = *reinterpret_cast<const bf16x8*>(wl + tap * (CI * W_ROW) + s * 32);
    };
    fetch(0, fr[0]);
#pragma unroll
    for (int s = 0; s < CO / 16; ++s) {
      if (s + 1 < CO / 16) fetch(s + 1, fr[(s + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);  // keep the requests above the multiplies they are meant to overlap
      const Frags& f = fr[s & 1];         // tap index = 3 ky + kx
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.t[4], f.a00, acc[0][0], 0, 0, 0);  // W11
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.t[5], f.a00, acc[0][1], 0, 0, 0);  // W12
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.t[7], f.a00, acc[1][0], 0, 0, 0);  // W21
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.t[8], f.a00, acc[1][1], 0, 0, 0);  // W22
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.t[3], f.a01, acc[0][1], 0, 0, 0);  // W10
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.t[6], f.a01, acc[1][1], 0, 0, 0);  // W20
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.t[1], f.a10, acc[1][0], 0, 0, 0);  // W01
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.t[2], f.a10, acc[1][1], 0, 0, 0);  // W02
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.t[0], f.a11, acc[1][1], 0, 0, 0);  // W00
      __builtin_amdgcn_sched_barrier(0);
    }
    // The products are formed TRANSPOSED (tap matrix as the A operand, dy pixels as B), so the C/D map reads: col = lane & 31 =
    // position, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) = ci — four consecutive ci per register quad, one 8-byte LDS write per
    // quad into patch[position][c][ci] (as 2-byte writes from the untransposed map the patch cost more LDS time than the
    // fragment reads: 64 ds_write_b16 per wave and dy row).
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          const f32x2 lo = {acc[a][c][4 * g], acc[a][c][4 * g + 1]}, hi = {acc[a][c][4 * g + 2], acc[a][c][4 * g + 3]};
          const uint2 v = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2)),   // v_cvt_pk_bf16_f32, RNE
                                     __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2)));
          *reinterpret_cast<uint2*>(patch + frow * PATCH_ROW + (c * CI + 8 * g + 4 * fh) * 2) = v;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own patch writes, before its lanes read each other's
      unsigned short* drow = dxb + ((long long)(2 * i + a) * (2 * WOUT) + 2 * j0) * ld_dx;
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // 32 positions x 8 chunks of 16 bytes over 64 lanes
        const int id = q * 64 + lane, pos = id >> 3, k = id & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(patch + pos * PATCH_ROW + k * 16);
        *reinterpret_cast<uint4*>(drow + (long long)(2 * pos + (k >> 2)) * ld_dx + (k & 3) * 8) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // patch reads done before it is overwritten
    }
  }
}

}  // namespace

extern "C" int hd_conv3x3s2_dgrad_supported(int B, int Hout, int Wout, int Cin, int Cout) {
  return B >= 1 && Hout >= 1 && Hout <= 4096 && Wout == WOUT && Cin == CI && Cout == CO && (long long)B * Hout <= 1 << 20;
}

extern "C" hd_status hd_conv3x3s2_dgrad(const void* dy, int64_t dy_row_stride, const void* w, int B, int Hout, int Wout, int Cin,
                                        int Cout, void* dx, int64_t dx_row_stride, hd_stream stream) {
  if (!dy || !w || !dx) return HD_ERR_NULL;
  if (!hd_conv3x3s2_dgrad_supported(B, Hout, Wout, Cin, Cout)) return HD_ERR_SHAPE;
  if (dy_row_stride == 0) dy_row_stride = Cout;
  if (dx_row_stride == 0) dx_row_stride = Cin;
  if (dy_row_stride < Cout || dx_row_stride < Cin || dy_row_stride % 8 || dx_row_stride % 8) return HD_ERR_SHAPE;
  if (!hd_aligned16(dy) || !hd_aligned16(w) || !hd_aligned16(dx)) return HD_ERR_ALIGN;
  // ~256 workgroups (one per CU: 134 KB of LDS), each a whole number of rows of one image; a walk pays one halo row
  int per_img = 256 / B;
  if (per_img < 1) per_img = 1;
  if (per_img > Hout) per_img = Hout;
  const int rows = (Hout + per_img - 1) / per_img;
  per_img = (Hout + rows - 1) / rows;
  (void)hd_launch(conv3x3s2_dgrad_c32_kernel, dim3((unsigned)(B * per_img)), dim3(NTHREAD), (size_t)LDS_BYTES, (hipStream_t)stream,
                  static_cast<const unsigned short*>(dy), (long long)dy_row_stride, static_cast<const unsigned short*>(w),
                  static_cast<unsigned short*>(dx), (long long)dx_row_stride, Hout, rows, per_img);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}
