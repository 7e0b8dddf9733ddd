// conv3x3s2_fwd.hip — forward of the FIRST stride-2 3x3 convolution of the YOLOv5 backbone (32 -> 64 channels, 320^2 -> 160^2 at
// 640^2) with the following BatchNorm's statistics in the epilogue: hd_conv3x3s2_fwd.
//
// Like its data gradient (conv3x3s2_dgrad.hip) the layer is memory-bound (15 GFLOP against 105 MB in and 52 MB out) and ran on
// the library (47 us) followed by a separate statistics pass over the 52 MB it had just written (12 us).  Here:
//   * a workgroup (5 waves = the 5 x 32 output positions of a row) walks down `rows_per_wg` output rows of one image; LDS
//     holds the three input rows of the current output row, each split into its even and odd pixels (entry stride 80 bytes),
//     so that tap kx of 32 consecutive output pixels is 32 consecutive entries of one plane (2j-1 -> odd[j], 2j -> even[j],
//     2j+1 -> odd[j+1]; odd[0] is the zero pixel left of the image) and the A fragments of 16 lanes fall on 16 bank groups;
//   * the next two input rows travel in registers while the current output row is multiplied (every input row is read from
//     memory once, + one halo row per walk); rows above / below the image are zeros;
//   * the weight stays in LDS as it lies in memory ([co][ky][kx][ci] = [64][288], rows padded to 592 bytes): k-step s of the
//     reduction is (tap s / 2, input channels 16 (s % 2) ...), the natural B fragment;
//   * the output row leaves through a wave-private patch as 16-byte stores (4 KB contiguous per wave), and the statistics of the
//     ROUNDED bf16 values are folded through LDS into one replica of the BatchNorm slot (as hd_pointwise_conv does).
#include "hd_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int kStatReplicas = 16;  // == hd_bn_act_replicas(), checked on the host
constexpr int CI = 32, CO = 64, WOUT = 160, WIN = 2 * WOUT;
constexpr int NWAVE = WOUT / 32, NTHREAD = NWAVE * 64;
constexpr int EN = CI * 2 + 16;                      // bytes per pixel entry of a plane
constexpr int ODD_OFF = WOUT * EN;                   // even plane: 160 entries; odd plane: 161 (entry 0 = pixel -1 = zero)
constexpr int ROW_BYTES = ODD_OFF + (WOUT + 1) * EN;
constexpr int KTOT = 9 * CI;                         // 288
constexpr int W_ROW = KTOT * 2 + 16;                 // 592
constexpr int W_BYTES = CO * W_ROW;
constexpr int PATCH_ROW = CO * 2 + 16, PATCH_BYTES = 32 * PATCH_ROW;
constexpr int LDS_BYTES = W_BYTES + 3 * ROW_BYTES + NWAVE * PATCH_BYTES + 2 * CO * (int)sizeof(float);
constexpr int ROW_CHUNKS = WIN * CI / 8;             // 16-byte chunks per input row (1280)
constexpr int LD_IT = 2 * ROW_CHUNKS / NTHREAD;      // two rows per fetch (8)
static_assert(ROW_CHUNKS % NTHREAD == 0, "a fetch instruction stays within one row");

template <bool STATS>
__global__ __launch_bounds__(NTHREAD) void conv3x3s2_fwd_c32_kernel(const unsigned short* __restrict__ x, long long ld_x,
                                                                     const unsigned short* __restrict__ w,
                                                                     unsigned short* __restrict__ y, long long ld_y, int H,
                                                                     int rows_per_wg, int per_img, float* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [weight | ring of 3 input rows | wave patches | fold]
  unsigned char* s_w = smem;
  unsigned char* s_ring = smem + W_BYTES;
  float* s_st = reinterpret_cast<float*>(s_ring + 3 * ROW_BYTES + NWAVE * PATCH_BYTES);  // [2][CO]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 31, fh = lane >> 5;
  unsigned char* patch = s_ring + 3 * ROW_BYTES + wave * PATCH_BYTES;
  const int b = blockIdx.x / per_img;
  const int r0 = (blockIdx.x % per_img) * rows_per_wg;
  const int r1 = r0 + rows_per_wg < H ? r0 + rows_per_wg : H;  // output rows [r0, r1)
  const int HIN = 2 * H;
  const unsigned short* xb = x + (long long)b * HIN * WIN * ld_x;
  unsigned short* yb = y + (long long)b * H * WOUT * ld_y;

  u32x4 pre[LD_IT];
  auto gload = [&](u32x4 (&pre)[LD_IT], int row0, int first, int last) {  // fetch instructions [first, last) of input rows row0, row0 + 1
#pragma unroll
    for (int p = 0; p < LD_IT; ++p) {
      if (p < first || p >= last) continue;
      const int id = p * NTHREAD + tid, row = row0 + id / ROW_CHUNKS, cid = id % ROW_CHUNKS;
      if (row >= 0 && row < HIN)  // rows outside the image are zeros (uniform per instruction)
        pre[p] = *reinterpret_cast<const u32x4*>(xb + ((long long)row * WIN + (cid >> 2)) * ld_x + (cid & 3) * 8);
      else
        pre[p] = u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto slot_of = [&](int row) { return s_ring + ((row + 3) % 3) * ROW_BYTES; };
  auto lstore = [&](const u32x4 (&pre)[LD_IT], int row0, int first, int last) {  // fetch instructions [first, last) of the two rows
#pragma unroll
    for (int p = 0; p < LD_IT; ++p) {
      if (p < first || p >= last) continue;
      const int id = p * NTHREAD + tid, row = row0 + id / ROW_CHUNKS, cid = id % ROW_CHUNKS;
      const int px = cid >> 2, c = cid & 3;
      unsigned char* dst = slot_of(row) + ((px & 1) ? ODD_OFF + ((px + 1) >> 1) * EN : (px >> 1) * EN) + c * 16;
      *reinterpret_cast<u32x4*>(dst) = pre[p];
    }
  };

  // Prologue: EVERY load of the start-up — the weight (8 chunks per thread), input rows 2 r0 - 1, 2 r0 and 2 r0 + 1 — is issued
  // before the first LDS store.  (As a load -> store loop the weight alone was eight L2 round trips per workgroup, and the rows
  // two more: ~10 us of a 44 us launch; the same finding as in the streaming 1x1 kernels)
  constexpr int WCH = CO * (KTOT / 8), WIT = (WCH + NTHREAD - 1) / NTHREAD;
  u32x4 wreg[WIT], pre_b[LD_IT];
#pragma unroll
  for (int p = 0; p < WIT; ++p) {
    const int id = p * NTHREAD + tid < WCH ? p * NTHREAD + tid : WCH - 1;  // threads past the end re-read the last chunk (never stored)
    wreg[p] = *reinterpret_cast<const u32x4*>(w + (long long)(id / (KTOT / 8)) * KTOT + (id % (KTOT / 8)) * 8);
  }
  gload(pre, 2 * r0 - 1, 0, LD_IT);        // rows 2 r0 - 1, 2 r0
  gload(pre_b, 2 * r0 + 1, 0, LD_IT / 2);  // row 2 r0 + 1 completes output row r0
  if (tid < 3 * 4)    // odd-plane entry 0 of each ring row: the zero pixel left of the image (never overwritten)
    *reinterpret_cast<u32x4*>(s_ring + (tid >> 2) * ROW_BYTES + ODD_OFF + (tid & 3) * 16) = u32x4{0u, 0u, 0u, 0u};
  if (STATS)
    for (int i = tid; i < 2 * CO; i += NTHREAD) s_st[i] = 0.0f;
#pragma unroll
  for (int p = 0; p < WIT; ++p) {  // the weight as it lies: [co][288] -> rows of 592 bytes
    const int id = p * NTHREAD + tid;
    if (id < WCH) *reinterpret_cast<u32x4*>(s_w + (id / (KTOT / 8)) * W_ROW + (id % (KTOT / 8)) * 16) = wreg[p];
  }
  lstore(pre, 2 * r0 - 1, 0, LD_IT);
  lstore(pre_b, 2 * r0 + 1, 0, LD_IT / 2);
  if (r0 + 1 < r1) gload(pre, 2 * r0 + 2, 0, LD_IT);  // rows 2 r0 + 2, 2 r0 + 3 travel while row r0 is multiplied

  const int j0 = wave * 32;
  float s_sum[2] = {0.0f, 0.0f}, s_sq[2] = {0.0f, 0.0f};
  // Invariant at the top of iteration i: rows 2i - 1, 2i, 2i + 1 are in the ring (stored, not yet fenced); `pre` holds rows
  // 2i + 2 and 2i + 3 (requested one iteration ago) unless i is the walk's last row.
  for (int i = r0; i < r1; ++i) {
    __syncthreads();  // the three rows of this output row are visible
    f32x16 acc[2];
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[jn][r] = 0.0f;
    const unsigned char* wl = s_w + frow * W_ROW + fh * 16;
    const unsigned char* rows[3] = {slot_of(2 * i - 1) + fh * 16 + (j0 + frow) * EN, slot_of(2 * i) + fh * 16 + (j0 + frow) * EN,
                                    slot_of(2 * i + 1) + fh * 16 + (j0 + frow) * EN};
    // k-step s = (tap s / 2 = 3 ky + kx, input channels 16 (s % 2) ..): the fragments of k-step s + PD are requested before the
    // MFMAs of k-step s issue.  Left to the compiler the loop was read, wait, multiply per fragment — one LDS round trip per MFMA,
    // 36 per output row and wave, which (not memory) was what the first version of this kernel spent its time on (41 us).
    constexpr int KS = 2 * 9, PD = 4;
    bf16x8 fa[PD + 1], f0[PD + 1], f1[PD + 1];
    auto fetch = [&](const int s, const int slot) {
      const int tap = s >> 1, ky = tap / 3, kx = tap % 3, h2 = s & 1;
      fa[slot] = *reinterpret_cast<const bf16x8*>(rows[ky] + (kx == 1 ? 0 : ODD_OFF) + (kx == 2 ? EN : 0) + h2 * 32);
      f0[slot] = *reinterpret_cast<const bf16x8*>(wl + s * 32);
      f1[slot] = *reinterpret_cast<const bf16x8*>(wl + 32 * W_ROW + s * 32);
    };
#pragma unroll
    for (int s = 0; s < PD; ++s) fetch(s, s);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + PD < KS) fetch(s + PD, (s + PD) % (PD + 1));
      __builtin_amdgcn_sched_barrier(0);  // keep the requests above the multiplies they are meant to overlap
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s % (PD + 1)], f0[s % (PD + 1)], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s % (PD + 1)], f1[s % (PD + 1)], acc[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();  // every wave has read rows 2i - 1 and 2i: their slots take rows 2i + 2 and 2i + 3
    if (i + 1 < r1) {
      lstore(pre, 2 * i + 2, 0, LD_IT);
      if (i + 2 < r1) gload(pre, 2 * i + 4, 0, LD_IT);
    }
    // C/D map: col = lane & 31 (co), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (position)  ->  patch[position][co]
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pos = (r & 3) + 8 * (r >> 2) + 4 * fh;
        const unsigned short v = __builtin_bit_cast(unsigned short, (__bf16)acc[jn][r]);  // v_cvt_pk_bf16_f32, RNE
        *reinterpret_cast<unsigned short*>(patch + pos * PATCH_ROW + (jn * 32 + frow) * 2) = v;
        if (STATS) {
          const float f = __uint_as_float((unsigned)v << 16);
          s_sum[jn] += f;
          s_sq[jn] = fmaf(f, f, s_sq[jn]);
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own patch writes, before its lanes read each other's
    unsigned short* drow = yb + ((long long)i * WOUT + j0) * ld_y;
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // 32 positions x 8 chunks of 16 bytes over 64 lanes
      const int id = q * 64 + lane, pos = id >> 3, k = id & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(patch + pos * PATCH_ROW + k * 16);
      *reinterpret_cast<uint4*>(drow + (long long)pos * ld_y + k * 8) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // patch reads done before it is overwritten
  }

  if (STATS) {
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
      const float s = s_sum[jn] + __shfl_xor(s_sum[jn], 32);
      const float q = s_sq[jn] + __shfl_xor(s_sq[jn], 32);
      if (fh == 0) {
        atomicAdd(&s_st[jn * 32 + frow], s);  // five waves per address
        atomicAdd(&s_st[CO + jn * 32 + frow], q);
      }
    }
    __syncthreads();
    float* dst = stats + (size_t)(blockIdx.x % kStatReplicas) * 2 * CO;
    for (int i = tid; i < 2 * CO; i += NTHREAD) atomicAdd(&dst[i], s_st[i]);
  }
}

}  // namespace

extern "C" int hd_conv3x3s2_fwd_supported(int B, int Hout, int Wout, int Cin, int Cout) {
  return B >= 1 && Hout >= 1 && Hout <= 4096 && Wout == WOUT && Cin == CI && Cout == CO && (long long)B * Hout <= 1 << 20;
}

extern "C" hd_status hd_conv3x3s2_fwd(const void* x, int64_t x_row_stride, const void* w, int B, int Hout, int Wout, int Cin,
                                      int Cout, void* y, int64_t y_row_stride, float* stats, hd_stream stream) {
  if (!x || !w || !y) return HD_ERR_NULL;
  if (!hd_conv3x3s2_fwd_supported(B, Hout, Wout, Cin, Cout)) return HD_ERR_SHAPE;
  if (x_row_stride == 0) x_row_stride = Cin;
  if (y_row_stride == 0) y_row_stride = Cout;
  if (x_row_stride < Cin || y_row_stride < Cout || x_row_stride % 8 || y_row_stride % 8) return HD_ERR_SHAPE;
  if (!hd_aligned16(x) || !hd_aligned16(w) || !hd_aligned16(y)) return HD_ERR_ALIGN;
  if (stats && hd_bn_act_replicas() != kStatReplicas) return HD_ERR_SHAPE;
  int per_img = 256 / B;  // ~256 workgroups (one per CU: 135 KB of LDS), each a whole number of output rows of one image
  if (per_img < 1) per_img = 1;
  if (per_img > Hout) per_img = Hout;
  const int rows = (Hout + per_img - 1) / per_img;
  per_img = (Hout + rows - 1) / rows;
  const auto* xp = static_cast<const unsigned short*>(x);
  const auto* wp = static_cast<const unsigned short*>(w);
  auto* yp = static_cast<unsigned short*>(y);
  if (stats)
    (void)hd_launch(conv3x3s2_fwd_c32_kernel<true>, dim3((unsigned)(B * per_img)), dim3(NTHREAD), (size_t)LDS_BYTES,
                    (hipStream_t)stream, xp, (long long)x_row_stride, wp, yp, (long long)y_row_stride, Hout, rows, per_img, stats);
  else
    (void)hd_launch(conv3x3s2_fwd_c32_kernel<false>, dim3((unsigned)(B * per_img)), dim3(NTHREAD), (size_t)LDS_BYTES,
                    (hipStream_t)stream, xp, (long long)x_row_stride, wp, yp, (long long)y_row_stride, Hout, rows, per_img, stats);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}
