// conv3x3_c32.hip — 3x3 / stride 1 / pad 1 convolution of 32 -> 32 channels on 160-pixel rows (the bottleneck 3x3 of the first C3
// block of YOLOv5s at 640^2), reached through hd_conv3x3_fwd (conv3x3.hip), whose tap kernel needs Cout in multiples of 64 and
// W <= 104.  With the flipped, transposed weights the same kernel is the layer's data gradient, as everywhere.
//
// 7.5 GFLOP against 26 MB in and 26 MB out: memory-bound; the library took 27 us (+ a statistics pass) forward and 33 us + a 9 us
// zero-fill for the data gradient.  The row walk of the first stride-2 layer's kernels (conv3x3s2_fwd.hip), stride 1:
//   * a workgroup (5 waves = the 5 x 32 positions of a row) walks down `rows_per_wg` rows of one image with a ring of three input
//     rows in LDS (162 entries of 80 bytes: entries 0 and 161 are the zero pixels left and right of the image, so tap kx of output
//     pixel j is entry j + kx) and the next row in registers; rows above / below the image are zeros;
//   * the weight stays in LDS as it lies ([co][ky][kx][ci] = [32][288], rows padded to 592 bytes) and is the B operand as it lies;
//   * 70 KB of LDS: two workgroups per CU;
//   * the output row leaves through a wave-private patch as 16-byte stores; statistics of the rounded values as in
//     hd_pointwise_conv.
#include "hd_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int kStatReplicas = 16;  // == hd_bn_act_replicas(), checked on the host
constexpr int CI = 32, CO = 32, WD = 160;
constexpr int NWAVE = WD / 32, NTHREAD = NWAVE * 64;
constexpr int EN = CI * 2 + 16;                 // bytes per pixel entry
constexpr int ROW_BYTES = (WD + 2) * EN;        // zero pixel | 160 pixels | zero pixel
constexpr int KTOT = 9 * CI;
constexpr int W_ROW = KTOT * 2 + 16;            // 592
constexpr int W_BYTES = CO * W_ROW;
constexpr int PATCH_ROW = CO * 2 + 16, PATCH_BYTES = 32 * PATCH_ROW;
constexpr int LDS_BYTES = W_BYTES + 3 * ROW_BYTES + NWAVE * PATCH_BYTES + 2 * CO * (int)sizeof(float);
constexpr int ROW_CHUNKS = WD * CI / 8;         // 16-byte chunks per row (640)
constexpr int LD_IT = ROW_CHUNKS / NTHREAD;     // 2
static_assert(ROW_CHUNKS % NTHREAD == 0, "row chunks per thread");

template <bool STATS>
__global__ __launch_bounds__(NTHREAD, 2) void conv3x3_c32_kernel(const unsigned short* __restrict__ x, long long ld_x,
                                                                  const unsigned short* __restrict__ w,
                                                                  unsigned short* __restrict__ y, long long ld_y, int H,
                                                                  int rows_per_wg, int per_img, float* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [weight | ring of 3 rows | wave patches | fold]
  unsigned char* s_w = smem;
  unsigned char* s_ring = smem + W_BYTES;
  float* s_st = reinterpret_cast<float*>(s_ring + 3 * ROW_BYTES + NWAVE * PATCH_BYTES);  // [2][CO]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 31, fh = lane >> 5;
  unsigned char* patch = s_ring + 3 * ROW_BYTES + wave * PATCH_BYTES;
  const int b = blockIdx.x / per_img;
  const int r0 = (blockIdx.x % per_img) * rows_per_wg;
  const int r1 = r0 + rows_per_wg < H ? r0 + rows_per_wg : H;  // output rows [r0, r1)
  const unsigned short* xb = x + (long long)b * H * WD * ld_x;
  unsigned short* yb = y + (long long)b * H * WD * ld_y;

  u32x4 pre[LD_IT];
  auto gload = [&](u32x4 (&pre)[LD_IT], int row) {  // rows outside the image are zeros (uniform branch)
#pragma unroll
    for (int p = 0; p < LD_IT; ++p) {
      const int id = p * NTHREAD + tid;
      if (row >= 0 && row < H)
        pre[p] = *reinterpret_cast<const u32x4*>(xb + ((long long)row * WD + (id >> 2)) * ld_x + (id & 3) * 8);
      else
        pre[p] = u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto slot_of = [&](int row) { return s_ring + ((row + 3) % 3) * ROW_BYTES; };
  auto lstore = [&](const u32x4 (&pre)[LD_IT], int row) {
    unsigned char* dst = slot_of(row) + EN;  // pixel p is entry p + 1
#pragma unroll
    for (int p = 0; p < LD_IT; ++p) {
      const int id = p * NTHREAD + tid;
      *reinterpret_cast<u32x4*>(dst + (id >> 2) * EN + (id & 3) * 16) = pre[p];
    }
  };

  // Prologue: the weight's chunks (4 per thread) and rows r0 - 1, r0 are all requested before the first LDS store (a load -> store
  // loop is one L2 round trip per chunk: see conv3x3s2_fwd.hip).
  constexpr int WCH = CO * (KTOT / 8), WIT = (WCH + NTHREAD - 1) / NTHREAD;
  u32x4 wreg[WIT], pre_b[LD_IT];
#pragma unroll
  for (int p = 0; p < WIT; ++p) {
    const int id = p * NTHREAD + tid < WCH ? p * NTHREAD + tid : WCH - 1;  // threads past the end re-read the last chunk (never stored)
    wreg[p] = *reinterpret_cast<const u32x4*>(w + (long long)(id / (KTOT / 8)) * KTOT + (id % (KTOT / 8)) * 8);
  }
  gload(pre_b, r0 - 1);
  gload(pre, r0);
  if (tid < 3 * 2 * 4) {  // the zero pixels left and right of each ring row (never overwritten)
    const int slot = tid / 8, side = (tid >> 2) & 1, c = tid & 3;
    *reinterpret_cast<u32x4*>(s_ring + slot * ROW_BYTES + (side ? (WD + 1) * EN : 0) + c * 16) = u32x4{0u, 0u, 0u, 0u};
  }
  if (STATS)
    for (int i = tid; i < 2 * CO; i += NTHREAD) s_st[i] = 0.0f;
#pragma unroll
  for (int p = 0; p < WIT; ++p) {  // the weight as it lies: [co][288] -> rows of 592 bytes
    const int id = p * NTHREAD + tid;
    if (id < WCH) *reinterpret_cast<u32x4*>(s_w + (id / (KTOT / 8)) * W_ROW + (id % (KTOT / 8)) * 16) = wreg[p];
  }
  lstore(pre_b, r0 - 1);
  lstore(pre, r0);
  gload(pre, r0 + 1);  // stored inside the first iteration

  const int j0 = wave * 32;
  float s_sum = 0.0f, s_sq = 0.0f;
  // Invariant at the top of iteration i: rows i - 1 and i are in the ring, `pre` holds row i + 1.
  for (int i = r0; i < r1; ++i) {
    lstore(pre, i + 1);    // into the slot row i - 2 left: its last reader passed the barrier at the end of iteration i - 1
    if (i + 1 < r1) gload(pre, i + 2);
    __syncthreads();  // the three rows of this output row are visible
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const unsigned char* wl = s_w + frow * W_ROW + fh * 16;
    const unsigned char* rows[3] = {slot_of(i - 1) + fh * 16 + (j0 + frow) * EN, slot_of(i) + fh * 16 + (j0 + frow) * EN,
                                    slot_of(i + 1) + fh * 16 + (j0 + frow) * EN};
    // k-step s = (tap s / 2 = 3 ky + kx, input channels 16 (s % 2) ..); fragments of k-step s + PD requested ahead of the MFMA of s
    constexpr int KS = 2 * 9, PD = 4;
    bf16x8 fa[PD + 1], fb[PD + 1];
    auto fetch = [&](const int s, const int slot) {
      const int tap = s >> 1, ky = tap / 3, kx = tap % 3, h2 = s & 1;
      fa[slot] = *reinterpret_cast<const bf16x8*>(rows[ky] + kx * EN + h2 * 32);  // pixel j - 1 + kx = entry j + kx
      fb[slot] = *reinterpret_cast<const bf16x8*>(wl + s * 32);
    };
#pragma unroll
    for (int s = 0; s < PD; ++s) fetch(s, s);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + PD < KS) fetch(s + PD, (s + PD) % (PD + 1));
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s % (PD + 1)], fb[s % (PD + 1)], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();  // every wave has read row i - 1: its slot takes row i + 2 in the next iteration
    // C/D map: col = lane & 31 (co), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (position)  ->  patch[position][co]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pos = (r & 3) + 8 * (r >> 2) + 4 * fh;
      const unsigned short v = __builtin_bit_cast(unsigned short, (__bf16)acc[r]);  // v_cvt_pk_bf16_f32, RNE
      *reinterpret_cast<unsigned short*>(patch + pos * PATCH_ROW + frow * 2) = v;
      if (STATS) {
        const float f = __uint_as_float((unsigned)v << 16);
        s_sum += f;
        s_sq = fmaf(f, f, s_sq);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own patch writes, before its lanes read each other's
    unsigned short* drow = yb + ((long long)i * WD + j0) * ld_y;
#pragma unroll
    for (int q = 0; q < 2; ++q) {  // 32 positions x 4 chunks of 16 bytes over 64 lanes
      const int id = q * 64 + lane, pos = id >> 2, k = id & 3;
      const uint4 v = *reinterpret_cast<const uint4*>(patch + pos * PATCH_ROW + k * 16);
      *reinterpret_cast<uint4*>(drow + (long long)pos * ld_y + k * 8) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // patch reads done before it is overwritten
  }

  if (STATS) {
    const float s = s_sum + __shfl_xor(s_sum, 32);
    const float q = s_sq + __shfl_xor(s_sq, 32);
    if (fh == 0) {
      atomicAdd(&s_st[frow], s);  // five waves per address
      atomicAdd(&s_st[CO + frow], q);
    }
    __syncthreads();
    float* dst = stats + (size_t)(blockIdx.x % kStatReplicas) * 2 * CO;
    for (int i = tid; i < 2 * CO; i += NTHREAD) atomicAdd(&dst[i], s_st[i]);
  }
}

}  // namespace

int hd_conv3x3_c32_supported(int B, int H, int W, int Cin, int Cout) {
  return B >= 1 && H >= 1 && H <= 4096 && W == WD && Cin == CI && Cout == CO && (long long)B * H <= 1 << 20;
}

hd_status hd_conv3x3_c32(const void* x, int64_t ldx, const void* w, int B, int H, void* y, int64_t ldy, float* stats, hipStream_t s) {
  int per_img = 512 / B;  // ~512 workgroups (two per CU: 70 KB of LDS each), each a whole number of rows of one image
  if (per_img < 1) per_img = 1;
  if (per_img > H) per_img = H;
  const int rows = (H + per_img - 1) / per_img;
  per_img = (H + rows - 1) / rows;
  const auto* xp = static_cast<const unsigned short*>(x);
  const auto* wp = static_cast<const unsigned short*>(w);
  auto* yp = static_cast<unsigned short*>(y);
  if (stats)
    (void)hd_launch(conv3x3_c32_kernel<true>, dim3((unsigned)(B * per_img)), dim3(NTHREAD), (size_t)LDS_BYTES, s, xp, (long long)ldx, wp,
                    yp, (long long)ldy, H, rows, per_img, stats);
  else
    (void)hd_launch(conv3x3_c32_kernel<false>, dim3((unsigned)(B * per_img)), dim3(NTHREAD), (size_t)LDS_BYTES, s, xp, (long long)ldx,
                    wp, yp, (long long)ldy, H, rows, per_img, stats);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}
