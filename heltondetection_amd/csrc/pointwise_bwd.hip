// pointwise_bwd.hip — the WHOLE backward of a 1x1 convolution on channels-last rows in one launch:
//     dX[R, Cin]    = dY[R, Cout] . W[Cout, Cin]                (the data gradient: reduction over Cout)
//     dW[Cout, Cin] = dY[R, Cout]^T . X[R, Cin]                 (the weight gradient: reduction over the rows)
// from ONE staging of dY (hd_pointwise_conv_dgrad and hd_pointwise_wgrad_partial each read dY, 26 + 31
// launches per YOLOv5s step, and the weight gradient's float32 slab moved as many bytes as its operands).
//
// Shape of the kernel = the streaming data-gradient kernel of pointwise_conv.hip (pointwise_stream_kernel<.., TRB>): a
// workgroup owns one BN-wide column tile of Cin, keeps W[:, tile] (as it lies: [Cout][BN] rows) in LDS for the whole launch,
// and its four waves never meet at a barrier inside the loop — each wave pulls blocks of 32 TM rows of dY coalesced into
// registers (the NEXT block in flight while the current one is multiplied), stages them in a wave-private swizzled LDS area,
// multiplies dX = dY . W through row fragments of that area and transposes the result through a wave-private patch so that
// the stores are 16 bytes per lane along full rows.  New here:
//   * the block's X rows [32 TM, BN] travel with it (registers -> a wave-private padded tile);
//   * dW[:, tile] += dY^T . X from the SAME staged dY rows, read a second time as LDS transpose fragments
//     (ds_read_b64_tr_b16: the reduction index of this product is the row index of both operands).  One swizzle serves both
//     kinds of read: 16 consecutive rows at one 16-byte chunk land on 16 bank groups (row fragments, ds_read_b128), and four
//     consecutive rows at four consecutive chunks land on the four 64-byte quarters of a bank row (transpose fragments);
//   * every wave keeps its [Cout, BN] float32 partial of dW in registers over ALL its row blocks (Cout / 32 x TN
//     accumulator tiles: 128 registers for 256 x 32 and 128 x 64); at the end the four waves' partials are summed
//     through LDS and ONE partial per workgroup goes to the slab [workgroups per column tile][Cout][Cin] with 16-byte
//     stores.  The slab is (workgroups) x Cout x BN x 4 bytes — 4-8 MB per launch whatever the layer, against
//     slices x Cout x Cin x 4 of the separate weight-gradient kernel (12.8 MB for 6400 x 256 x 256, 33 MB for 25600 x 256 x 256);
//   * ACC: the gradient that reaches x through a shortcut is added to dX in the epilogue (as hd_pointwise_conv_dgrad_add);
//   * workgroups past the main grid sum the row-sliced partials of the PREVIOUS weight-gradient launch (hd_wgrad_job).
// Rows past R read as zeros (buffer range check), so they add nothing to dW; their dX stores are dropped by the same check.
#include <hip/hip_bf16.h>

#include <cstdlib>
#include <type_traits>

#include "hd_bn_fuse.h"
#include "hd_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// [rows][KR] bf16, 16-byte chunks XOR-swizzled by a function of the row that serves row reads AND transpose reads.
// Address arithmetic: every access of the kernel is `lane part ^ step constant + step constant` (the identities below), so the
// lane parts are computed once per kernel and an access costs at most one v_xor — the straightforward `off(row, chunk)` per
// access was ~12 VALU instructions each, 460 per 32-row block of the 256-deep variant against 32 MFMAs (PMC: SQ_INSTS_VALU).
template <int KR>
struct Stage {
  static constexpr int CPR = KR / 8;
  static constexpr int RPI = 64 / CPR;  // rows covered by one 64-lane staging instruction
  __host__ __device__ static constexpr int f(int row) {
    return CPR >= 16 ? (((row & 3) << 2) | ((row >> 2) & 3)) : (CPR == 8 ? ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)) : ((row >> 2) & 3));
  }
  __host__ __device__ static constexpr int off(int row, int chunk) { return row * (KR * 2) + ((chunk ^ f(row)) << 4); }
  // (1) staging: row = p RPI + rl (rl < RPI)            ->  f(row) = f(rl) ^ f(p RPI)
  // (2) fragments: row + 16 k, row + 32 i                ->  f unchanged
  // (3) transpose fragments: row + 4 with row % 8 < 4    ->  taken as a second lane part (no identity needed)
  static constexpr bool identities() {
    for (int row = 0; row < 128; ++row) {
      if (f(row + 16) != f(row) || f(row + 32) != f(row)) return false;
      if (f(row) != (f(row % RPI) ^ f(row - row % RPI))) return false;
      if (f(row) >= CPR) return false;
    }
    return true;
  }
};
static_assert(Stage<32>::identities() && Stage<64>::identities() && Stage<128>::identities() && Stage<256>::identities(),
              "the staging swizzle must be XOR-separable as pointwise_bwd_kernel assumes");

// transpose fragment (8 consecutive rows of one 32-column tile) of a Stage<KR> image from the two lane parts of its address
__device__ __forceinline__ bf16x8 tr_fragment_at(const unsigned char* lo_addr, const unsigned char* hi_addr) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lo_addr));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(hi_addr));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// the same of a plain padded image (row stride `stride` bytes): the W slab and the X tile
__device__ __forceinline__ bf16x8 tr_fragment(const unsigned char* tile, int stride, int col0, int m_base, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const unsigned char* a = tile + (m_base + 8 * (g >> 1) + q) * stride + (col0 + 16 * (g & 1) + 4 * p) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * stride));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

constexpr int kRideBlocks = 8;                               // reduce blocks of a riding job per extra workgroup
constexpr int kRideLds = kRideBlocks * 16 * 17 * 16;         // their LDS (float4 [kRideBlocks][16][17])

template <int KR, int TN, int TM>
struct BwdLds {
  static constexpr int BN = TN * 32;
  static constexpr int SLAB_ROW = BN * 2 + 64;            // W[:, tile] rows, padded for the transpose reads
  static constexpr int SLAB_BYTES = KR * SLAB_ROW;
  static constexpr int XROW = BN * 2 + (BN == 32 ? 0 : 64);  // 64-byte rows need no padding (four rows = four bank quarters)
  static constexpr int STAGE_BYTES = 32 * TM * KR * 2;
  static constexpr int PATCH_ROW = BN * 2 + 16;
  static constexpr int PATCH_BYTES = 32 * PATCH_ROW;
  static constexpr int AREA0 = STAGE_BYTES > PATCH_BYTES ? STAGE_BYTES : PATCH_BYTES;  // dY stage, then the output patch
  static constexpr int XTILE_BYTES = 32 * TM * XROW;
  static constexpr int WAVE_BYTES = AREA0 + XTILE_BYTES;
  static constexpr int LOOP_BYTES = SLAB_BYTES + 4 * WAVE_BYTES;
  static constexpr int FOLD_BYTES = 4 * KR * BN * 4;       // the four waves' partials of dW[:, tile], summed at the end
  static constexpr int MAIN = LOOP_BYTES > FOLD_BYTES ? LOOP_BYTES : FOLD_BYTES;
  static constexpr int TOTAL = MAIN > kRideLds ? MAIN : kRideLds;
};

template <int KR, int TN, int TM, bool ACC>
__global__ __launch_bounds__(256, (KR <= 64 ? 2 : 1)) void pointwise_bwd_kernel(const unsigned short* __restrict__ dY, long long ld_dy,
                                                            const unsigned short* __restrict__ Wt,   // [KR][N] forward weight
                                                            const unsigned short* __restrict__ X, long long ld_x,
                                                            unsigned short* __restrict__ dX, long long ld_dx, int M, int N,
                                                            const unsigned short* __restrict__ radd, long long ld_r,
                                                            float* __restrict__ slab, int G, const hdfuse::ReduceJob job,
                                                            int kr_real) {
  // kr_real <= KR: the weight has only kr_real rows (output channels); the rest of the reduction is padding — dY's columns
  // past kr_real must be ZERO (hd_head_conv_bwd: the 255 -> 256 Detect convolution), the slab rows read as zeros and the
  // partials of dW keep kr_real rows per slice
  using L = BwdLds<KR, TN, TM>;
  constexpr int BN = TN * 32, KS = KR / 16, CPR = KR / 8, NTL = KR / 32;
  constexpr int NLD = TM * KS;           // 16-byte dY loads per lane per row block
  constexpr int XCPR = BN / 8;           // 16-byte chunks per X row
  constexpr int NLX = TM * BN / 16;      // 16-byte X loads per lane per row block (32 TM rows x XCPR chunks over 64 lanes)
  static_assert(64 % CPR == 0 && 64 % XCPR == 0, "a lane's chunk index must not depend on the load index");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [W slab | 4 wave areas], later the fold area
  if ((int)blockIdx.x >= G) {  // the slice sum of the previous weight-gradient launch rides along (hd_wgrad_job)
    // kRideBlocks reduce blocks per workgroup: this launch holds one or two workgroups per CU, so the loads in flight have to
    // come from each thread (hd_bn_fuse.h: reduce_blocks_multi)
    const long long first = ((long long)blockIdx.x - G) * kRideBlocks;
    if (first < hdfuse::reduce_blocks(job.count))
      hdfuse::reduce_blocks_multi<kRideBlocks>(job, first, reinterpret_cast<float4(*)[16][17]>(smem));
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: row-block offsets stay in SGPRs
  const int frow = lane & 31, fh = lane >> 5;
  const int NT = N / BN;
  const int lb = (G % 8 == 0) ? ((int)blockIdx.x % 8) * (G / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
  const int nt = lb % NT, n0 = nt * BN;                    // G is a multiple of NT: one column tile per workgroup
  const int slice = lb / NT;                               // this workgroup among those of its column tile
  const int gw = slice * 4 + wave, GW = (G / NT) * 4;
  const int RT = (M + 32 * TM - 1) / (32 * TM);
  const int my = gw < RT ? (RT - 1 - gw) / GW + 1 : 0;

  // Every global access of the loop is a buffer instruction: a descriptor per tensor whose range ends with row M - 1, the lane's
  // byte offset in ONE VGPR for the whole kernel, the row block's offset in an SGPR.  The hardware range check returns zeros for
  // rows past M (so they add nothing to dW) and drops their dX stores: no clamps, no predicates, every memory instruction of a
  // block unconditional — the compiler counts them and waits for exactly the block it needs (pointwise_conv.hip, `block`), and
  // the 64-bit address arithmetic per load (~6 VALU instructions, 100 per block) is gone.  The host checks rows * stride < 2 GiB.
  constexpr int RPX = 64 / XCPR;  // rows of the X / dX / addend tile covered by one 64-lane instruction
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)dY, 0, (int)((long long)M * ld_dy * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)((long long)M * ld_x * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dx = __builtin_amdgcn_make_buffer_rsrc((void*)dX, 0, (int)((long long)M * ld_dx * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void*)(ACC ? radd : dY), 0, (int)((long long)M * (ACC ? ld_r : ld_dy) * 2), 0x00020000);
  const int vo_dy = (lane / CPR) * (int)ld_dy * 2 + (lane % CPR) * 16;
  const int vo_x = (lane / XCPR) * (int)ld_x * 2 + (lane % XCPR) * 16 + n0 * 2;
  const int vo_dx = (lane / XCPR) * (int)ld_dx * 2 + (lane % XCPR) * 16 + n0 * 2;
  const int vo_r = (lane / XCPR) * (int)ld_r * 2 + (lane % XCPR) * 16 + n0 * 2;
  const int sdy = (int)ld_dy * 2, sx = (int)ld_x * 2, sdx = (int)ld_dx * 2, sr = (int)ld_r * 2;  // row strides in bytes

  // D row blocks travel in registers ahead of the one being multiplied (one workgroup per CU for the deep reductions, so only
  // the wave's own requests cover the memory latency): block it + D is requested as soon as block it has left its registers.
  constexpr int D = KR >= 128 ? 2 : 1;  // KR <= 64 runs two waves per SIMD: one block of lookahead, half the registers
  u32x4 nxt[D][NLD], nxx[D][NLX];
  auto load_rows = [&](int rt, u32x4 (&by)[NLD], u32x4 (&bx)[NLX]) {
    const int r0 = rt * (32 * TM);
#pragma unroll
    for (int p = 0; p < NLD; ++p) by[p] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, vo_dy, (r0 + p * (64 / CPR)) * sdy, 0);
#pragma unroll
    for (int p = 0; p < NLX; ++p) bx[p] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo_x, (r0 + p * RPX) * sx, 0);
  };
  if (my > 0) load_rows(gw, nxt[0], nxx[0]);  // in flight while the weight slab is staged
  if (D > 1 && my > 1) load_rows(gw + GW, nxt[D - 1], nxx[D - 1]);
  {
    constexpr int NCH = KR * XCPR, WIT = (NCH + 255) / 256;  // all of a thread's slab loads before its first LDS store
    u32x4 wreg[WIT];
#pragma unroll
    for (int p = 0; p < WIT; ++p) {
      const int id = p * 256 + tid;
      wreg[p] = u32x4{0u, 0u, 0u, 0u};
      if ((NCH % 256 == 0 || id < NCH) && id / XCPR < kr_real)
        wreg[p] = *reinterpret_cast<const u32x4*>(Wt + (long long)(id / XCPR) * N + n0 + (id % XCPR) * 8);
    }
#pragma unroll
    for (int p = 0; p < WIT; ++p) {
      const int id = p * 256 + tid;
      if (NCH % 256 == 0 || id < NCH) *reinterpret_cast<u32x4*>(smem + (id / XCPR) * L::SLAB_ROW + (id % XCPR) * 16) = wreg[p];
    }
  }
  __syncthreads();

  unsigned char* area = smem + L::SLAB_BYTES + wave * L::WAVE_BYTES;  // wave-private: dY stage, then the output patch
  unsigned char* xt = area + L::AREA0;                                 // wave-private X tile
  // lane parts of the stage addresses (Stage<KR>): staging stores, row fragments, the two halves of a transpose fragment
  constexpr int RPI = Stage<KR>::RPI;
  const int st_base = Stage<KR>::off(lane / CPR, lane % CPR);
  const int ra_base = Stage<KR>::off(frow, fh);
  const int tr_g = lane >> 4, tr_q = (lane & 15) >> 2, tr_p = lane & 3;
  const int tr_row = 8 * (tr_g >> 1) + tr_q, tr_chunk = 2 * (tr_g & 1) + (tr_p >> 1), tr_within = (tr_p & 1) * 8;
  const int tr_lo = Stage<KR>::off(tr_row, tr_chunk) + tr_within;
  const int tr_hi = Stage<KR>::off(tr_row + 4, tr_chunk) + tr_within;
  const int xt_st = (lane / XCPR) * L::XROW + (lane % XCPR) * 16;   // X tile staging store of this lane (+ p RPX rows)
  f32x16 wacc[NTL][TN];
#pragma unroll
  for (int n = 0; n < NTL; ++n)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) wacc[n][j][r] = 0.0f;

  auto block = [&](const int it, u32x4 (&by)[NLD], u32x4 (&bx)[NLX]) {
    const int rt = gw + it * GW;
#pragma unroll
    for (int p = 0; p < NLD; ++p)  // rows p RPI + lane / CPR: identity (1)
      *reinterpret_cast<u32x4*>(area + (st_base ^ (Stage<KR>::f(p * RPI) << 4)) + p * RPI * (KR * 2)) = by[p];
#pragma unroll
    for (int p = 0; p < NLX; ++p) *reinterpret_cast<u32x4*>(xt + xt_st + p * RPX * L::XROW) = bx[p];
    load_rows(rt + D * GW, by, bx);   // past the wave's last block: out of range, nothing moves
    u32x4 racc[ACC ? TM : 1][ACC ? BN / 16 : 1];
    if (ACC) {  // requested now, consumed after the MFMA phase
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < BN / 16; ++q)
          racc[i][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, vo_r, ((rt * TM + i) * 32 + q * RPX) * sr, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS writes, before its lanes read each other's
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    // The two products as ONE software pipeline of KS + 2 TM NTL steps: the operand fragments of step u + PD are requested
    // before the MFMAs of step u are issued.  With one or two waves per SIMD nobody else covers an LDS round trip, and the
    // compiler's own schedule is read -> wait -> multiply per fragment (pointwise_conv.hip: 46 % of the wave cycles parked).
    //   steps 0 .. KS-1:        dX += dY[:, 16 s ..] . W[16 s .., tile]      row fragments of the stage x transpose fragments of the slab
    //   steps KS .. KS+T-1:     dW[32 n .., tile] += dY[16 ks .., 32 n ..]^T . X[16 ks .., tile]   transpose fragments of both
    constexpr int T = 2 * TM * NTL, S = KS + T, PD = 2;
    bf16x8 fa[PD + 1][TM], fb[PD + 1][TN], fy[PD + 1], fx[2 * TM][TN];
    auto fetch = [&](const int u) {
      const int slot = u % (PD + 1);
      if (u < KS) {
#pragma unroll
        for (int i = 0; i < TM; ++i)  // row 32 i + frow, chunk 2 u + fh: identity (2), and 2 u only has bits the lane part leaves alone
          fa[slot][i] = *reinterpret_cast<const bf16x8*>(area + (ra_base ^ (u << 5)) + i * 32 * (KR * 2));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[slot][j] = tr_fragment(smem, L::SLAB_ROW, j * 32, u * 16, lane);
      } else {
        const int t = u - KS, ks = t / NTL, n = t % NTL;
        if (n == 0) {
#pragma unroll
          for (int j = 0; j < TN; ++j) fx[ks][j] = tr_fragment(xt, L::XROW, j * 32, ks * 16, lane);
        }
        // rows 16 ks + ..., columns 32 n + ...: chunk 4 n + lane part -> ^ (n << 6); identity (2) for the rows
        fy[slot] = tr_fragment_at(area + (tr_lo ^ (n << 6)) + ks * 16 * (KR * 2), area + (tr_hi ^ (n << 6)) + ks * 16 * (KR * 2));
      }
    };
#pragma unroll
    for (int u = 0; u < PD; ++u) fetch(u);
#pragma unroll
    for (int u = 0; u < S; ++u) {
      if (u + PD < S) fetch(u + PD);
      __builtin_amdgcn_sched_barrier(0);  // keep the requests above the multiplies they are meant to overlap
      const int slot = u % (PD + 1);
      if (u < KS) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[slot][i], fb[slot][j], acc[i][j], 0, 0, 0);
      } else {
        const int t = u - KS, ks = t / NTL, n = t % NTL;
#pragma unroll
        for (int j = 0; j < TN; ++j) wacc[n][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[slot], fx[ks][j], wacc[n][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // stage reads done before the patch overwrites the area
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m0 = (rt * TM + i) * 32;
      // C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)  ->  patch[row][col] as bf16
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
          const unsigned short v = __builtin_bit_cast(unsigned short, (__bf16)acc[i][j][r]);  // v_cvt_pk_bf16_f32, RNE
          *reinterpret_cast<unsigned short*>(area + row * L::PATCH_ROW + (j * 32 + frow) * 2) = v;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < BN / 16; ++q) {  // 32 rows x BN/8 chunks of 16 bytes over 64 lanes
        const int id = q * 64 + lane, row = id / XCPR, c = id % XCPR;
        u32x4 v = *reinterpret_cast<const u32x4*>(area + row * L::PATCH_ROW + c * 16);
        if (ACC) {
          const u32x4 r = racc[i][q];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 f = {__uint_as_float(v[e] << 16) + __uint_as_float(r[e] << 16),
                             __uint_as_float(v[e] & 0xffff0000u) + __uint_as_float(r[e] & 0xffff0000u)};
            v[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
          }
        }
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_dx, vo_dx, (m0 + q * RPX) * sdx, 0);  // rows past M: dropped by the range check
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // patch reads done before it is overwritten
    }
  };
  {
    int it = 0;
    if (D > 1) {
      for (; it + 2 <= my; it += 2) {
        block(it, nxt[0], nxx[0]);
        block(it + 1, nxt[D - 1], nxx[D - 1]);
      }
      if (it < my) block(it, nxt[0], nxx[0]);
    } else {
      for (; it < my; ++it) block(it, nxt[0], nxx[0]);
    }
  }

  // ---- the four waves' partials of dW[:, tile] -> one partial of the workgroup -> slab[slice][KR][N]
  __syncthreads();  // every wave is past its last slab / stage read
  float* fold = reinterpret_cast<float*>(smem) + (size_t)wave * (KR * BN);
#pragma unroll
  for (int n = 0; n < NTL; ++n)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) fold[((n * TN + j) * 16 + r) * 64 + lane] = wacc[n][j][r];
  __syncthreads();
  const float* f0 = reinterpret_cast<const float*>(smem);
  float* out = slab + (size_t)slice * kr_real * N;
  for (int e = tid; e < kr_real * (BN / 4); e += 256) {
    const int row = e / (BN / 4), c = (e % (BN / 4)) * 4;        // dW row (output channel), first of 4 columns of the tile
    const int n = row >> 5, rr = row & 31;
    const int r = (rr & 3) + 4 * (rr >> 3), h = (rr >> 2) & 1;   // C/D map inverted: row = (r & 3) + 8 (r >> 2) + 4 h
    const int idx = ((n * TN + (c >> 5)) * 16 + r) * 64 + h * 32 + (c & 31);
    const float4 a = *reinterpret_cast<const float4*>(f0 + idx);
    const float4 b = *reinterpret_cast<const float4*>(f0 + KR * BN + idx);
    const float4 c2 = *reinterpret_cast<const float4*>(f0 + 2 * KR * BN + idx);
    const float4 d = *reinterpret_cast<const float4*>(f0 + 3 * KR * BN + idx);
    const float4 s = make_float4((a.x + b.x) + (c2.x + d.x), (a.y + b.y) + (c2.y + d.y), (a.z + b.z) + (c2.z + d.z),
                                 (a.w + b.w) + (c2.w + d.w));
    *reinterpret_cast<float4*>(out + (size_t)row * N + n0 + c) = s;
  }
}

struct BwdPlan {
  int tn, tm, bn, nt, grid, slices;
};

// Column tile and grid of a shape.  Register budget: the wave's dW partial is (Cout / 32) x TN accumulator tiles of 16
// registers, kept at <= 128 registers.  Grid: a multiple of 8 (XCD remap) and of the column-tile count; at most the
// resident cap (one or two workgroups per CU by LDS) and at most one row block per wave.
inline BwdPlan bwd_plan(int64_t R, int Cin, int Cout, int want_grid) {
  BwdPlan p{};
  int tn_max = Cout >= 256 ? 1 : 2;  // 64 x 128 and 32 x 128 tiles spill (256 + 256 registers with the addend in flight)
  int tn = tn_max;
  while (tn > 1 && Cin % (32 * tn)) tn >>= 1;
  p.tn = tn;
  p.tm = (Cout <= 32 && tn == 1) ? 2 : 1;  // 32-row blocks keep the 64-wide variants at two workgroups per CU (LDS and registers)
  p.bn = 32 * tn;
  p.nt = Cin / p.bn;
  const long long RT = (R + 32 * p.tm - 1) / (32 * p.tm);
  long long L = 8;
  while (L % p.nt) L += 8;
  long long g = (RT + 3) / 4 * p.nt;  // one row block per wave
  g = (g + L - 1) / L * L;
  // Main grid.  `two`: two workgroups fit a CU (LDS <= 80 KB: the 64- and 32-deep variants and 128 x 32).  Measured in the
  // YOLOv5s 640^2 bs16 step (profiles/r04_pointwise_bwd.md): the one-per-CU variants at 160-224 workgroups 5.55 ms/step, at 256
  // 5.63 — with every CU taken by a main workgroup the riding slice sum of the previous layer only starts when the first main
  // workgroups retire; at 192 a quarter of the CUs take the riders at once (and the slab is a quarter smaller).  The
  // two-per-CU variants: 128 / 256 equal, 512 slower (5.67: twice the partials to write and sum).
  const bool two = Cout <= 64 || (Cout == 128 && tn == 1);
  long long cap = want_grid > 0 ? want_grid : (two ? 256 : 192);
  if (cap < L) cap = L;
  if (g > cap) g = cap / L * L;
  p.grid = (int)g;
  p.slices = p.grid / p.nt;
  return p;
}

inline bool bwd_shape_ok(int64_t R, int Cin, int Cout) {
  if (!(R >= 1 && R <= 0x7fffffff)) return false;
  if (!(Cout == 32 || Cout == 64 || Cout == 128 || Cout == 256)) return false;
  if (Cin < 32 || Cin % 32 || Cin > 8192) return false;
  const BwdPlan p = bwd_plan(R, Cin, Cout, 0);
  return p.nt <= 32;
}

constexpr int g_bwd_grid_override = 0;   // the plan's own grid (round 4 swept it through an exported setter, removed in round 5)

}  // namespace

extern "C" int hd_pointwise_bwd_supported(int64_t R, int Cin, int Cout) { return bwd_shape_ok(R, Cin, Cout) ? 1 : 0; }

extern "C" size_t hd_pointwise_bwd_workspace_bytes(int64_t R, int Cin, int Cout) {
  if (!bwd_shape_ok(R, Cin, Cout)) return 0;
  const BwdPlan p = bwd_plan(R, Cin, Cout, g_bwd_grid_override);
  return (size_t)p.slices * Cin * Cout * sizeof(float);
}

// Cout: the reduction the kernel runs (32 / 64 / 128 / 256); cout_real <= Cout: the rows the weight and its gradient have
static hd_status bwd_launch(const void* dy, int64_t dy_row_stride, const void* w, const void* x, int64_t x_row_stride,
                            int64_t R, int Cin, int Cout, int cout_real, const void* add, int64_t add_row_stride, void* dx,
                            int64_t dx_row_stride, void* dw, void* workspace, size_t workspace_bytes,
                            const hd_wgrad_job* piggyback, hd_wgrad_job* job_out, hd_stream stream) {
  if (!dy || !w || !x || !dx || !dw || !workspace || !job_out) return HD_ERR_NULL;
  if (!bwd_shape_ok(R, Cin, Cout) || cout_real < 1 || cout_real > Cout) return HD_ERR_SHAPE;
  if (dy_row_stride == 0) dy_row_stride = Cout;
  if (x_row_stride == 0) x_row_stride = Cin;
  if (dx_row_stride == 0) dx_row_stride = Cin;
  if (add && add_row_stride == 0) add_row_stride = Cin;
  if (dy_row_stride < Cout || x_row_stride < Cin || dx_row_stride < Cin || dy_row_stride % 8 || x_row_stride % 8 || dx_row_stride % 8)
    return HD_ERR_SHAPE;
  if (add && (add_row_stride < Cin || add_row_stride % 8)) return HD_ERR_SHAPE;
  if (!hd_aligned16(dy) || !hd_aligned16(w) || !hd_aligned16(x) || !hd_aligned16(dx) || !hd_aligned16(dw) || !hd_aligned16(workspace) ||
      (add && !hd_aligned16(add)))
    return HD_ERR_ALIGN;
  // buffer instructions address rows with 32-bit byte offsets from the tensor's first element
  const int64_t lim = (int64_t)1 << 31;
  if (R * dy_row_stride * 2 >= lim || R * x_row_stride * 2 >= lim || R * dx_row_stride * 2 >= lim || (add && R * add_row_stride * 2 >= lim))
    return HD_ERR_SHAPE;
  const BwdPlan p = bwd_plan(R, Cin, Cout, g_bwd_grid_override);
  if (workspace_bytes < (size_t)p.slices * Cin * Cout * sizeof(float)) return HD_ERR_WORKSPACE;
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
  const unsigned extra = jv.slab ? (unsigned)((hdfuse::reduce_blocks(jv.count) + kRideBlocks - 1) / kRideBlocks) : 0u;
  hipStream_t s = (hipStream_t)stream;
  const auto* a = static_cast<const unsigned short*>(dy);
  const auto* b = static_cast<const unsigned short*>(w);
  const auto* c = static_cast<const unsigned short*>(x);
  const auto* r = static_cast<const unsigned short*>(add);
  auto* o = static_cast<unsigned short*>(dx);
  float* slab = static_cast<float*>(workspace);
  const int M = (int)R;
  const dim3 grid((unsigned)p.grid + extra);
#define HD_BWD_LAUNCH(KRR, TNN, TMM)                                                                                          \
  do {                                                                                                                        \
    const size_t lds = BwdLds<KRR, TNN, TMM>::TOTAL;                                                                          \
    if (add)                                                                                                                  \
      (void)hd_launch(pointwise_bwd_kernel<KRR, TNN, TMM, true>, grid, dim3(256), lds, s, a, (long long)dy_row_stride, b, c,  \
                      (long long)x_row_stride, o, (long long)dx_row_stride, M, Cin, r, (long long)add_row_stride, slab, p.grid, jv, \
                      cout_real);                                                                                              \
    else                                                                                                                      \
      (void)hd_launch(pointwise_bwd_kernel<KRR, TNN, TMM, false>, grid, dim3(256), lds, s, a, (long long)dy_row_stride, b, c, \
                      (long long)x_row_stride, o, (long long)dx_row_stride, M, Cin, r, 0ll, slab, p.grid, jv, cout_real);     \
  } while (0)
  if (Cout == 256) HD_BWD_LAUNCH(256, 1, 1);
  else if (Cout == 128) { if (p.tn == 2) HD_BWD_LAUNCH(128, 2, 1); else HD_BWD_LAUNCH(128, 1, 1); }
  else if (Cout == 64) { if (p.tn == 2) HD_BWD_LAUNCH(64, 2, 1); else HD_BWD_LAUNCH(64, 1, 1); }
  else { if (p.tn == 2) HD_BWD_LAUNCH(32, 2, 1); else HD_BWD_LAUNCH(32, 1, 2); }
#undef HD_BWD_LAUNCH
  HD_RETURN_IF_LAUNCH_FAILED();
  job_out->slab = slab;
  job_out->slices = p.slices;
  job_out->count = (int64_t)Cin * cout_real;
  job_out->dw = dw;
  return HD_OK;
}

extern "C" hd_status hd_pointwise_bwd(const void* dy, int64_t dy_row_stride, const void* w, const void* x, int64_t x_row_stride,
                                      int64_t R, int Cin, int Cout, const void* add, int64_t add_row_stride, void* dx,
                                      int64_t dx_row_stride, void* dw, void* workspace, size_t workspace_bytes,
                                      const hd_wgrad_job* piggyback, hd_wgrad_job* job_out, hd_stream stream) {
  return bwd_launch(dy, dy_row_stride, w, x, x_row_stride, R, Cin, Cout, Cout, add, add_row_stride, dx, dx_row_stride, dw, workspace,
                    workspace_bytes, piggyback, job_out, stream);
}

// ---- the Detect convolution's backward: Cout = na (5 + nc) is not a tile multiple (255 for 80 classes) ------------------------
// dY arrives as rows of `hd_head_conv_pitch(Cout)` bf16 columns whose padding columns are ZERO (the loss kernels write the
// padded gradient, ops.head_conv casts it); the kernel runs the padded reduction against weight rows that read as zeros past
// Cout, and the weight gradient keeps Cout rows.
extern "C" int hd_head_conv_pitch(int Cout) {
  if (Cout < 1) return 0;
  for (int p = 32; p <= 256; p <<= 1)
    if (Cout <= p) return p;
  return 0;
}

extern "C" int hd_head_conv_bwd_supported(int64_t R, int Cin, int Cout) {
  const int p = Cout >= 1 ? hd_head_conv_pitch(Cout) : 0;
  return p && bwd_shape_ok(R, Cin, p) ? 1 : 0;
}

extern "C" size_t hd_head_conv_bwd_workspace_bytes(int64_t R, int Cin, int Cout) {
  const int p = Cout >= 1 ? hd_head_conv_pitch(Cout) : 0;
  return p ? hd_pointwise_bwd_workspace_bytes(R, Cin, p) : 0;
}

extern "C" hd_status hd_head_conv_bwd(const void* dy, int64_t dy_row_stride, const void* w, const void* x, int64_t x_row_stride,
                                      int64_t R, int Cin, int Cout, void* dx, int64_t dx_row_stride, void* dw, void* workspace,
                                      size_t workspace_bytes, const hd_wgrad_job* piggyback, hd_wgrad_job* job_out,
                                      hd_stream stream) {
  const int p = Cout >= 1 ? hd_head_conv_pitch(Cout) : 0;
  if (!p) return HD_ERR_SHAPE;
  if (dy_row_stride == 0) dy_row_stride = p;
  return bwd_launch(dy, dy_row_stride, w, x, x_row_stride, R, Cin, p, Cout, nullptr, 0, dx, dx_row_stride, dw, workspace,
                    workspace_bytes, piggyback, job_out, stream);
}
