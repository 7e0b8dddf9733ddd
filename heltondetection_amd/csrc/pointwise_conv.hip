// pointwise_conv.hip — 1x1 convolution on channels-last rows as an MFMA GEMM, with the BatchNorm statistics of the
// fused BN+activation that follows it taken in the epilogue.
//
// Why hand-written.  34 of YOLOv5s' 57 convolutions are 1x1, stride 1: on channels-last tensors that is
//     Y[R, Cout] = X[R, Cin] . W[Cout, Cin]^T          (R = B*H*W rows)
// with R up to 409 600 and Cin, Cout <= 1024 — a skinny GEMM whose arithmetic intensity (<= Cout flops per input
// byte) keeps it on the HBM roofline, not the MFMA one.  The library kernels MIOpen picks for these shapes run at
// 0.2-0.8 of that roofline, and a separate statistics kernel then reads Y again.  Here one persistent launch streams X
// once, keeps W in LDS / L2, and folds sum(y), sum(y^2) per output channel (of the bf16-rounded y, the values the
// normalisation will read) into the replicated accumulators of csrc/bn_act.hip, which removes that layer's
// bn_stats_kernel launch and its pass over Y.
// The same kernel is the data gradient (dX[R, Cin] = dY[R, Cout] . Wt[Cin, Cout]^T with the transposed weights).
//
// Tiling (wave64, 4 waves per workgroup).  A workgroup owns a BM x BN output tile and walks K in BK = 64 (or 32) steps;
// both operands are K-contiguous, so a tile row is 128 (64) bytes and is staged global -> registers -> LDS in full
// 16-byte chunks, coalesced over whole rows.  LDS rows are XOR-swizzled at 16-byte granularity
// (chunk ^ (row / rows_per_256B) % chunks_per_row) so that both the ds_write_b128 of the staging pass and the
// ds_read_b128 of the fragment reads (16 consecutive rows, same chunk, per 16-lane group) are bank-conflict free.
// Each wave computes TM x TN tiles of v_mfma_f32_32x32x16_bf16 (lane l holds row/col l & 31, k = 8 (l >> 5) + j).
// The grid is persistent (<= 512 workgroups = 2 per CU): a workgroup loops over output tiles with the loads of its next
// (tile, k-step) in flight while it computes the current one from the other LDS buffer, so single-k-step shapes
// (Cin = 64) still overlap load and compute, and the statistics need one global atomic per channel per workgroup.
// Workgroup ids are remapped so that consecutive logical ids (the N-tiles of one row block, which share X) run on one
// XCD and hit in its L2.
#include <hip/hip_bf16.h>

#include "hd_bn_fuse.h"
#include <cstdlib>
#include <type_traits>

#include "hd_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int kStatReplicas = 16;  // == hd_bn_act_replicas(), checked on the host

template <int BK>
struct Tile {  // [rows][BK] bf16 LDS image, 16-byte chunks swizzled
  static constexpr int CPR = BK / 8;    // chunks per row
  static constexpr int RPL = 16 / CPR;  // rows per 256-byte bank line
  __device__ static __forceinline__ int off(int row, int chunk) {
    return row * (BK * 2) + ((chunk ^ ((row / RPL) % CPR)) << 4);
  }
};

__device__ __forceinline__ float bf16_bits_to_float(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ unsigned short float_to_bf16_bits(float f) {  // round to nearest even (finite inputs; NaN kept)
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

template <int BM, int BN, int WM, int WN, int BK, bool STATS>
__global__ __launch_bounds__(256) void pointwise_conv_kernel(const unsigned short* __restrict__ A, long long lda,
                                                             const unsigned short* __restrict__ Bw,
                                                             unsigned short* __restrict__ Cmat, long long ldc, int M, int N,
                                                             int K, float* __restrict__ stats) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  static_assert(TM >= 1 && TN >= 1, "tile too small for the wave grid");
  constexpr int CPR = BK / 8;
  constexpr int A_CHUNKS = BM * CPR, B_CHUNKS = BN * CPR;
  constexpr int A_IT = (A_CHUNKS + 255) / 256, B_IT = (B_CHUNKS + 255) / 256;
  constexpr int BUF_BYTES = (BM + BN) * BK * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 x [A tile | B tile]; reused for the statistics fold

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int MT = (M + BM - 1) / BM, NT = N / BN, KT = K / BK;
  const int G = gridDim.x;
  // consecutive logical ids on one XCD (hardware deals consecutive blockIdx round-robin over the 8 XCDs)
  const int lb = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
  const long long tiles = (long long)MT * NT;
  const int my_tiles = lb < tiles ? (int)((tiles - 1 - lb) / G) + 1 : 0;
  const int W = my_tiles * KT;

  uint4 ra[A_IT], rb[B_IT];
  auto gload = [&](int w) {
    const long long t = lb + (long long)(w / KT) * G;
    const int kt = w % KT;
    const int m0 = (int)(t / NT) * BM, n0 = (int)(t % NT) * BN;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int id = i * 256 + tid;
      if (A_CHUNKS % 256 == 0 || id < A_CHUNKS) {
        int m = m0 + id / CPR;
        m = m < M ? m : M - 1;  // rows past the end re-read the last row; their results are never stored
        ra[i] = *reinterpret_cast<const uint4*>(A + (long long)m * lda + kt * BK + (id % CPR) * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int id = i * 256 + tid;
      if (B_CHUNKS % 256 == 0 || id < B_CHUNKS)
        rb[i] = *reinterpret_cast<const uint4*>(Bw + (long long)(n0 + id / CPR) * K + kt * BK + (id % CPR) * 8);
    }
  };
  auto lstore = [&](int buf) {
    unsigned char* sa = smem + buf * BUF_BYTES;
    unsigned char* sb = sa + BM * BK * 2;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int id = i * 256 + tid;
      if (A_CHUNKS % 256 == 0 || id < A_CHUNKS) *reinterpret_cast<uint4*>(sa + Tile<BK>::off(id / CPR, id % CPR)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int id = i * 256 + tid;
      if (B_CHUNKS % 256 == 0 || id < B_CHUNKS) *reinterpret_cast<uint4*>(sb + Tile<BK>::off(id / CPR, id % CPR)) = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  float s_sum[TN], s_sq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) s_sum[j] = s_sq[j] = 0.0f;

  const int frow = lane & 31, fh = lane >> 5;
  if (W > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  for (int w = 0; w < W; ++w) {
    if (w + 1 < W) gload(w + 1);
    const unsigned char* sa = smem + (w & 1) * BUF_BYTES;
    const unsigned char* sb = sa + BM * BK * 2;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        a[i] = *reinterpret_cast<const bf16x8*>(sa + Tile<BK>::off((wm * TM + i) * 32 + frow, 2 * s + fh));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = *reinterpret_cast<const bf16x8*>(sb + Tile<BK>::off((wn * TN + j) * 32 + frow, 2 * s + fh));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (w % KT == KT - 1) {  // the tile is complete: write it (C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
      const long long t = lb + (long long)(w / KT) * G;
      const int m0 = (int)(t / NT) * BM, n0 = (int)(t % NT) * BN;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + (wn * TN + j) * 32 + frow;
          const int mb = m0 + (wm * TM + i) * 32 + 4 * fh;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mb + (r & 3) + 8 * (r >> 2);
            if (m < M) {
              const unsigned short v = float_to_bf16_bits(acc[i][j][r]);
              Cmat[(long long)m * ldc + n] = v;
              if (STATS) {
                const float f = bf16_bits_to_float(v);
                s_sum[j] += f;
                s_sq[j] = fmaf(f, f, s_sq[j]);
              }
            }
            acc[i][j][r] = 0.0f;
          }
        }
    }
    if (w + 1 < W) lstore((w + 1) & 1);
    __syncthreads();
  }

  if (STATS && my_tiles > 0) {
    // every tile of this workgroup has the same N-tile (G is a multiple of NT), so the lane's columns never changed
    float* s_st = reinterpret_cast<float*>(smem);  // [2][BN]
    for (int i = tid; i < 2 * BN; i += 256) s_st[i] = 0.0f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float a = s_sum[j] + __shfl_xor(s_sum[j], 32);
      const float q = s_sq[j] + __shfl_xor(s_sq[j], 32);
      if (fh == 0) {
        const int c = (wn * TN + j) * 32 + frow;
        atomicAdd(&s_st[c], a);  // WM waves per address
        atomicAdd(&s_st[BN + c], q);
      }
    }
    __syncthreads();
    const int n0 = (int)(lb % NT) * BN;
    float* dst = stats + (size_t)(blockIdx.x % kStatReplicas) * 2 * N;
    for (int i = tid; i < 2 * BN; i += 256) atomicAdd(&dst[(i / BN) * N + n0 + (i % BN)], s_st[i]);
  }
}

// ---- streaming form: Cin <= 256 --------------------------------------------------------------------------------
// With a short reduction the weight slab of a workgroup's column tile (BN x K bf16 <= 32 KB) stays in LDS for the whole
// launch and the waves never meet at a barrier again: each wave owns 32 TM rows of X, pulls them straight into MFMA
// operand registers (lane l: row l & 31, 16-byte chunk 2 s + (l >> 5) of k-step s — the natural fragment, 32 bytes per
// row per instruction), has the NEXT row block's loads in flight while it multiplies the current one, and transposes
// its result through a wave-private LDS patch so that the global stores are 16 bytes per lane along full rows.
template <int K>
struct Slab {  // [rows][K] bf16, 16-byte chunks swizzled so that 16 consecutive rows at one chunk index hit 16 bank groups
  static constexpr int CPR = K / 8;
  __device__ static __forceinline__ int off(int row, int chunk) {
    const int f = CPR >= 16 ? (row & 15) : (row / (16 / CPR)) % CPR;
    return row * (K * 2) + ((chunk ^ f) << 4);
  }
};

typedef __attribute__((ext_vector_type(4))) short s16x4;

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

// TRB: the weight operand is given [K][N] (the forward weight, seen from the data gradient): its slab keeps that
// layout (rows of BN columns, padded by 64 bytes) and the B fragments are LDS transpose reads instead of row reads.
template <int K, int TN, int TM, bool TRB = false>
struct StreamLds {
  static constexpr int BN = TN * 32;
  static constexpr int SLAB_ROW_T = BN * 2 + 64;
  static constexpr int SLAB_BYTES = TRB ? K * SLAB_ROW_T : BN * K * 2;
  static constexpr int PATCH_ROW = BN * 2 + 16;  // +16: the two half-waves' rows (4 apart) land 16 banks apart
  static constexpr int STAGE_BYTES = 32 * TM * K * 2, PATCH_BYTES = 32 * PATCH_ROW;
  static constexpr int WAVE_BYTES = STAGE_BYTES > PATCH_BYTES ? STAGE_BYTES : PATCH_BYTES;  // one area, two uses in turn
  static constexpr int TOTAL = SLAB_BYTES + 4 * WAVE_BYTES;
};

// BNB: the A operand is dy of a fused BatchNorm + activation, formed on the fly from d(out) (= A, possibly in two places)
// and the BatchNorm's input y (hd_bn_fuse.h): each 16-byte chunk is transformed between its global load and its LDS
// store, with the 8 channels' coefficients in registers (a lane's chunk index is fixed: K / 8 divides 64).
// ACC: the result is added to the rows of `radd` (bf16, row stride ld_r) before it is stored — the data gradient of a layer
// whose input also feeds a shortcut: dx = dy . w + d(shortcut) in one pass (float32 add of the two bf16 values, one rounding:
// what the separate bf16 add kernel of the autograd engine computes), instead of a launch that reads both and writes the sum.
// DEEP: the long-launch form (>= kDeepBlocks row blocks per wave: the 832^2 / 1280^2 levels): two row blocks of register
// lookahead with the loads issued behind the previous block's stores, every memory instruction of the steady-state loop
// unconditional, operand fragments requested two k-steps ahead.  The short form (DEEP = false) is the round-2 kernel: on the
// 640^2 pyramid a wave owns 1-4 blocks, the steady-state loop never runs and the deeper prologue costs 1-8 us per launch.
// HEAD: the Detect convolution (hd_head_conv_fwd): the weight has n_real <= N rows (the rest of the column range is padding and
// reads as zeros), `stats` is the float32 bias [n_real], and the result leaves as FLOAT32 rows (Cmat: float*, ldc in floats)
// = accumulator + bias, unrounded; padding columns are written as zeros.  Reductions
// longer than the resident slab run as K-slices: ld_r is the weight's row stride (the slice starts at Bw), and with `radd`
// non-null the launch ADDS its product to what the previous slice left in y (no bias: `stats` null).
template <int K, int TN, int TM, bool STATS, bool TRB, bool BNB = false, bool ACC = false, bool DEEP = false, bool HEAD = false>
__global__ __launch_bounds__(256) void pointwise_stream_kernel(const unsigned short* __restrict__ A, long long lda,
                                                               const unsigned short* __restrict__ Bw,
                                                               unsigned short* __restrict__ Cmat, long long ldc, int M, int N,
                                                               float* __restrict__ stats, const hdfuse::BnBwd bn,
                                                               const unsigned short* __restrict__ radd, long long ld_r,
                                                               int n_real) {
  using L = StreamLds<K, TN, TM, TRB>;
  constexpr int BN = TN * 32, KS = K / 16, CPR = K / 8;
  constexpr int NLD = TM * KS;  // 16-byte loads per lane per row block: 32 TM rows x CPR chunks over 64 lanes
  static_assert(64 % CPR == 0, "a lane's chunk index must not depend on the load index");
  __shared__ float s_bn[BNB ? 4 * K : 1];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [slab | 4 wave areas]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 31, fh = lane >> 5;
  const int NT = N / BN, G = gridDim.x;
  const int lb = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
  const int nt = lb % NT, n0 = nt * BN;                    // G is a multiple of NT: one column tile per workgroup
  const int gw = (lb / NT) * 4 + wave, GW = (G / NT) * 4;  // this wave among the waves of its column tile
  const int RT = (M + 32 * TM - 1) / (32 * TM);            // row blocks of 32 TM rows
  const int my = gw < RT ? (RT - 1 - gw) / GW + 1 : 0;

  // coalesced: instruction p of a lane takes chunk (p * 64 + lane) of the row block, i.e. 64 / CPR whole rows at a time
  // D row blocks travel in registers ahead of the one being multiplied: 2 for the deep variants (K >= 128, one or two waves
  // per SIMD), whose loads are issued behind the previous block's stores (see `block` below); 1 for K <= 64 (4 waves per SIMD).
  constexpr int D = (DEEP && K >= 128 && !BNB) ? 2 : 1;
  u32x4 nxt[D][NLD], nyy[BNB ? NLD : 1];
  hdfuse::Place pa;
  pa.base = A + (lane % CPR) * 8;
  pa.ld = lda;
  if (BNB) pa.set(A, lda, bn, (lane % CPR) * 8);
  auto load_rows = [&](int rt, u32x4 (&dst)[NLD]) {
#pragma unroll
    for (int p = 0; p < NLD; ++p) {
      const int id = p * 64 + lane;
      int m = rt * (32 * TM) + id / CPR;
      m = m < M ? m : M - 1;  // rows past the end re-read the last row; their results are never stored
      dst[p] = *reinterpret_cast<const u32x4*>(pa.base + (long long)m * pa.ld);
      if (BNB) nyy[p] = *reinterpret_cast<const u32x4*>(bn.y + (long long)m * bn.ld_y + (id % CPR) * 8);
    }
  };
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < my) load_rows(gw + d * GW, nxt[d]);  // in flight while the weight slab is staged
  if (BNB) hdfuse::fill_coefs(bn, 0, K, s_bn, false);
  {
    // the slab's BN x K / 8 chunks: ALL of a thread's loads are issued before its first LDS store (a load -> store loop waits
    // out one L2 round trip per chunk: 8 of them, ~5 us, for the 32 KB slabs — half the launch on the 20 x 20 / 40 x 40 levels)
    constexpr int NCH = BN * CPR, WIT = (NCH + 255) / 256;
    u32x4 wreg[WIT];
#pragma unroll
    for (int p = 0; p < WIT; ++p) {
      const int id = p * 256 + tid;
      if (NCH % 256 == 0 || id < NCH) {
        if (TRB) {
          const int row = id / (BN / 8), c = id % (BN / 8);
          wreg[p] = *reinterpret_cast<const u32x4*>(Bw + (long long)row * N + n0 + c * 8);
        } else {
          const int row = id / CPR, c = id % CPR;
          if (HEAD && n0 + row >= n_real) wreg[p] = u32x4{0u, 0u, 0u, 0u};
          else wreg[p] = *reinterpret_cast<const u32x4*>(Bw + (long long)(n0 + row) * (HEAD ? ld_r : (long long)K) + c * 8);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < WIT; ++p) {
      const int id = p * 256 + tid;
      if (NCH % 256 == 0 || id < NCH) {
        if (TRB) {
          const int row = id / (BN / 8), c = id % (BN / 8);
          *reinterpret_cast<u32x4*>(smem + row * L::SLAB_ROW_T + c * 16) = wreg[p];
        } else {
          const int row = id / CPR, c = id % CPR;
          *reinterpret_cast<u32x4*>(smem + Slab<K>::off(row, c)) = wreg[p];
        }
      }
    }
  }
  __syncthreads();

  constexpr int HEAD_PATCH = 32 * (BN * 4 + 32);   // HEAD: the patch holds float32 rows
  constexpr int WAVE_BYTES = (HEAD && HEAD_PATCH > L::WAVE_BYTES) ? HEAD_PATCH : L::WAVE_BYTES;
  unsigned char* area = smem + L::SLAB_BYTES + wave * WAVE_BYTES;  // wave-private: row-block stage, then output patch
  float s_sum[TN], s_sq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) s_sum[j] = s_sq[j] = 0.0f;
  hdfuse::Coef8 cf;
  if (BNB) cf.load(s_bn, K, (lane % CPR) * 8);  // behind the barrier above
  float hbias[HEAD ? TN : 1];
  if (HEAD) {
#pragma unroll
    for (int j = 0; j < TN; ++j) hbias[j] = (stats && n0 + j * 32 + frow < n_real) ? stats[n0 + j * 32 + frow] : 0.0f;
  }

  // One row block.  GUARD: rows past M exist in this block (or it is one of the wave's last blocks): stores and the next
  // loads are conditional.  Without it every memory instruction of the block is issued unconditionally, which is what lets the
  // compiler count them: behind a conditional store or load it must assume the instruction was NOT issued, and its
  // s_waitcnt vmcnt(n) for the oldest loads then also drains everything younger (measured: the wait at the head of a block
  // included the acknowledgement of the previous block's stores, or — with two blocks of lookahead — the loads issued a moment
  // before; 46 % of the wave cycles of the K = 256 variants were parked, PMC on 692 224 x 256 x 256).
  auto block = [&](auto guard_tag, const int it, u32x4 (&buf)[NLD]) {
    constexpr bool GUARD = decltype(guard_tag)::value;
    const int rt = gw + it * GW;
#pragma unroll
    for (int p = 0; p < NLD; ++p) {
      const int id = p * 64 + lane;
      *reinterpret_cast<u32x4*>(area + Slab<K>::off(id / CPR, id % CPR)) = BNB ? hdfuse::chunk(buf[p], nyy[p], cf, bn.act) : buf[p];
    }
    if (D == 1 && it + D < my) load_rows(rt + D * GW, buf);
    // ACC: the addend's chunks of this row block are requested now and consumed after the MFMA phase (a load -> add -> store
    // chain inside the store loop made the fused form slower than the data gradient plus a separate add launch)
    u32x4 racc[ACC ? TM : 1][ACC ? BN / 16 : 1];
    if (ACC) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < BN / 16; ++q) {
          const int id = q * 64 + lane, row = id / (BN / 8), c = id % (BN / 8);
          int m = (rt * TM + i) * 32 + row;
          m = m < M ? m : M - 1;
          racc[i][q] = *reinterpret_cast<const u32x4*>(radd + (long long)m * ld_r + n0 + c * 8);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS writes, before its lanes read each other's
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    // The operand fragments of k-step s + PD are requested before the MFMAs of k-step s are issued.  With one or two waves
    // per SIMD nobody else covers an LDS round trip: the compiler's own schedule (read, wait, multiply, per fragment) left the
    // deep variants 46 % of their wave cycles parked in s_waitcnt and the matrix pipe 20 % busy (PMC, 692 224 x 256 x 256).
    constexpr int PD = (DEEP && K >= 128) ? 2 : 0;
    if constexpr (PD == 0) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        bf16x8 a[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(area + Slab<K>::off(i * 32 + frow, 2 * s + fh));
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const bf16x8 b = TRB ? tr_fragment(smem, L::SLAB_ROW_T, j * 32, s * 16, lane)
                               : *reinterpret_cast<const bf16x8*>(smem + Slab<K>::off(j * 32 + frow, 2 * s + fh));
#pragma unroll
          for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b, acc[i][j], 0, 0, 0);
        }
      }
    } else {
      bf16x8 fa[PD + 1][TM], fb[PD + 1][TN];
      auto fetch = [&](int s, bf16x8 (&a)[TM], bf16x8 (&b)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(area + Slab<K>::off(i * 32 + frow, 2 * s + fh));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[j] = TRB ? tr_fragment(smem, L::SLAB_ROW_T, j * 32, s * 16, lane)
                     : *reinterpret_cast<const bf16x8*>(smem + Slab<K>::off(j * 32 + frow, 2 * s + fh));
      };
#pragma unroll
      for (int s = 0; s < PD; ++s) fetch(s, fa[s], fb[s]);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s + PD < KS) fetch(s + PD, fa[(s + PD) % (PD + 1)], fb[(s + PD) % (PD + 1)]);
        __builtin_amdgcn_sched_barrier(0);  // keep the requests above the multiplies they are meant to overlap
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int i = 0; i < TM; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s % (PD + 1)][i], fb[s % (PD + 1)][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // stage reads done before the patch overwrites the area
    if constexpr (HEAD) {
      // float32 rows through the wave's patch ([32][BN] floats, rows padded by 32 bytes: the two half-waves' rows, 4 apart, land
      // 32 banks apart), then 16 bytes per lane along full rows — as the bf16 epilogue below.  (Stored straight from the
      // accumulator layout, 4 bytes per lane, the 80 x 80 level took 87 us for 130 MB.)
      constexpr int PR = BN * 4 + 32;
      float* Cf = reinterpret_cast<float*>(Cmat);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m0 = (rt * TM + i) * 32;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
            *reinterpret_cast<float*>(area + row * PR + (j * 32 + frow) * 4) = acc[i][j][r] + hbias[j];
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < BN / 8; ++q) {  // 32 rows x BN/4 chunks of 16 bytes over 64 lanes
          const int id = q * 64 + lane, row = id / (BN / 4), c = id % (BN / 4);
          float4 v = *reinterpret_cast<const float4*>(area + row * PR + c * 16);
          if (!GUARD || m0 + row < M) {
            float4* dst = reinterpret_cast<float4*>(Cf + (long long)(m0 + row) * ldc + n0 + c * 4);
            if (radd) {
              const float4 o = *dst;
              v = make_float4(v.x + o.x, v.y + o.y, v.z + o.z, v.w + o.w);
            }
            *dst = v;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // patch reads done before it is overwritten
      }
    } else {
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
          if (STATS && m0 + row < M) {
            const float f = bf16_bits_to_float(v);
            s_sum[j] += f;
            s_sq[j] = fmaf(f, f, s_sq[j]);
          }
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < BN / 16; ++q) {  // 32 rows x BN/8 chunks of 16 bytes over 64 lanes
        const int id = q * 64 + lane, row = id / (BN / 8), c = id % (BN / 8);
        uint4 v = *reinterpret_cast<const uint4*>(area + row * L::PATCH_ROW + c * 16);
        if (!GUARD || m0 + row < M) {
          if (ACC) {
            const u32x4 r = racc[i][q];
            const unsigned vv[4] = {v.x, v.y, v.z, v.w}, rr[4] = {r[0], r[1], r[2], r[3]};
            unsigned oo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
              typedef float f32x2 __attribute__((ext_vector_type(2)));
              const f32x2 f = {__uint_as_float(vv[e] << 16) + __uint_as_float(rr[e] << 16),
                               __uint_as_float(vv[e] & 0xffff0000u) + __uint_as_float(rr[e] & 0xffff0000u)};
              oo[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
            }
            v = make_uint4(oo[0], oo[1], oo[2], oo[3]);
          }
          *reinterpret_cast<uint4*>(Cmat + (long long)(m0 + row) * ldc + n0 + c * 8) = v;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // patch reads done before it is overwritten
    }
    }
    // D == 2: the loads of row block it + 2 are issued BEHIND this block's stores, so that everything older than the loads a
    // block waits for is a full block old (vmcnt retires in issue order); rows past M are clamped inside load_rows
    if (D > 1 && (!GUARD || it + D < my)) load_rows(rt + D * GW, buf);
  };
  {
    const std::integral_constant<bool, true> guarded;
    const std::integral_constant<bool, false> plain;
    int it = 0;
    if (D > 1) {
      const int full_rt = M / (32 * TM);                                   // row blocks without rows past M
      const int my_full = gw < full_rt ? (full_rt - 1 - gw) / GW + 1 : 0;  // this wave's share of them (<= my)
      for (; it + 2 <= my_full; it += 2) {
        block(plain, it, nxt[0]);
        block(plain, it + 1, nxt[D - 1]);
      }
    }
    for (; it < my; ++it) {  // D == 1: every block; D == 2: the last one or two, and the ragged one
      if (D == 1 || (it & 1) == 0) block(guarded, it, nxt[0]);
      else block(guarded, it, nxt[D - 1]);
    }
  }

  if (STATS) {
    __syncthreads();  // every wave is past its last slab read: the slab's first bytes become the fold area
    float* s_st = reinterpret_cast<float*>(smem);  // [2][BN]
    for (int i = tid; i < 2 * BN; i += 256) s_st[i] = 0.0f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float a = s_sum[j] + __shfl_xor(s_sum[j], 32);
      const float q = s_sq[j] + __shfl_xor(s_sq[j], 32);
      if (fh == 0) {
        atomicAdd(&s_st[j * 32 + frow], a);
        atomicAdd(&s_st[BN + j * 32 + frow], q);
      }
    }
    __syncthreads();
    float* dst = stats + (size_t)(blockIdx.x % kStatReplicas) * 2 * N;
    for (int i = tid; i < 2 * BN; i += 256) atomicAdd(&dst[(i / BN) * N + n0 + (i % BN)], s_st[i]);
  }
}

constexpr long long kDeepBlocks = 6;  // row blocks per wave from which the DEEP form of the streaming kernel is launched

template <int K, int TN, int TM, bool TRB>
hd_status launch_stream(const void* A, int64_t lda, const void* Bw, void* Cmat, int64_t ldc, int M, int N, float* stats,
                        hipStream_t s, const hdfuse::BnBwd* bn = nullptr, const void* radd = nullptr, int64_t ld_r = 0) {
  constexpr int BN = TN * 32;
  const int NT = N / BN;
  const long long RT = ((long long)M + 32 * TM - 1) / (32 * TM);
  const size_t lds = StreamLds<K, TN, TM, TRB>::TOTAL;
  // persistent cap = what is resident at once: two workgroups per CU when two fit in LDS, else one (K = 256: 96-112 KB;
  // 512 workgroups then ran as two rounds, each paying its slab load and first-block latency)
  const long long cap = lds > 80 * 1024 ? 256 : 512;
  long long g = (RT + 3) / 4 * NT;  // one row block per wave, or the persistent cap
  long long L = 8;                  // the grid is a multiple of 8 (XCD remap) and of NT (one column tile per workgroup)
  while (L % NT) L += 8;
  g = (g + L - 1) / L * L;
  if (g > cap) g = cap / L * L;
  const auto* a = static_cast<const unsigned short*>(A);
  const auto* b = static_cast<const unsigned short*>(Bw);
  auto* c = static_cast<unsigned short*>(Cmat);
  const hdfuse::BnBwd none = {};
  const unsigned short* no_add = nullptr;
  // row blocks per wave of this launch: the deep form pays off from kDeepBlocks on (measured on the 640^2 and 832^2 pyramids)
  const bool deep = K >= 128 && !bn && RT * NT >= kDeepBlocks * g * 4;
  if (bn) {
    if constexpr (TRB)
      (void)hd_launch(pointwise_stream_kernel<K, TN, TM, false, true, true>, dim3((unsigned)g), dim3(256), lds, s, a, lda, b, c,
                      ldc, M, N, (float*)nullptr, *bn, no_add, 0ll, 0);
    else
      return HD_ERR_SHAPE;
  } else if (radd) {
    if constexpr (TRB) {
      const auto* r = static_cast<const unsigned short*>(radd);
      if (deep)
        (void)hd_launch(pointwise_stream_kernel<K, TN, TM, false, true, false, true, (K >= 128)>, dim3((unsigned)g), dim3(256), lds, s,
                        a, lda, b, c, ldc, M, N, (float*)nullptr, none, r, (long long)ld_r, 0);
      else
        (void)hd_launch(pointwise_stream_kernel<K, TN, TM, false, true, false, true>, dim3((unsigned)g), dim3(256), lds, s, a, lda,
                        b, c, ldc, M, N, (float*)nullptr, none, r, (long long)ld_r, 0);
    } else {
      return HD_ERR_SHAPE;
    }
  } else if (stats && !TRB) {
    if (deep)
      (void)hd_launch(pointwise_stream_kernel<K, TN, TM, true, false, false, false, (K >= 128)>, dim3((unsigned)g), dim3(256), lds, s,
                      a, lda, b, c, ldc, M, N, stats, none, no_add, 0ll, 0);
    else
      (void)hd_launch(pointwise_stream_kernel<K, TN, TM, true, false>, dim3((unsigned)g), dim3(256), lds, s, a, lda, b, c, ldc, M,
                      N, stats, none, no_add, 0ll, 0);
  } else {
    if (deep)
      (void)hd_launch(pointwise_stream_kernel<K, TN, TM, false, TRB, false, false, (K >= 128)>, dim3((unsigned)g), dim3(256), lds, s,
                      a, lda, b, c, ldc, M, N, (float*)nullptr, none, no_add, 0ll, 0);
    else
      (void)hd_launch(pointwise_stream_kernel<K, TN, TM, false, TRB>, dim3((unsigned)g), dim3(256), lds, s, a, lda, b, c, ldc, M,
                      N, (float*)nullptr, none, no_add, 0ll, 0);
  }
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

// -> true when the streaming form took the call
// Column tiles of one streaming launch: every row block is pulled through L2 once per column tile (the workgroups of one row
// range sit on one XCD), so wide outputs cost L2 bandwidth, not HBM; 32 covers ResNet50's 256 -> 1024 / 1024 <- 256 layers.
constexpr int kMaxColumnTiles = 32;
constexpr long long kWideRows = 40000;    // rows from which the K = 256 kernels take 128-wide column tiles (see try_stream)

template <int K, bool TRB = false>
bool try_stream(const void* A, int64_t lda, const void* Bw, void* Cmat, int64_t ldc, int M, int N, float* stats, hipStream_t s,
                hd_status* st, const hdfuse::BnBwd* bnb = nullptr, const void* radd = nullptr, int64_t ld_r = 0) {
  // K = 256: a 128-wide column tile needs a 64 KB slab (one workgroup per CU either way): its load is the start-up cost of the
  // launch, which loses on the 40 x 40 / 20 x 20 levels (+3-5 us) and wins once there are enough row blocks per workgroup to
  // amortise it — every row block is then pulled through L2 by N / 128 instead of N / 64 workgroups
  // ... and only for launches that take the deep form: a 128-wide tile in the short form was measured slower wherever it ran
  // (the 80 x 80 level of YOLOv5l: +0.4 ms per step): the wide launch has N / 128 column tiles on <= 256 workgroups
  const long long wide_blocks = ((long long)M + 31) / 32 * (N / 128);
  const bool wide = K > 128 && !bnb && M >= kWideRows && N % 128 == 0 && wide_blocks >= kDeepBlocks * 256 * 4;
  const int max_bn = (K <= 128 || wide) ? 128 : 64;
  const int bn = N % 128 == 0 && max_bn >= 128 ? 128 : (N % 64 == 0 ? 64 : 32);
  if (N / bn > kMaxColumnTiles) return false;
  if (bn == 128) {
    *st = launch_stream<K, 4, 1, TRB>(A, lda, Bw, Cmat, ldc, M, N, stats, s, bnb, radd, ld_r);
    return true;
  }
  if (bn == 64) {
    if constexpr (K <= 64) *st = launch_stream<K, 2, 2, TRB>(A, lda, Bw, Cmat, ldc, M, N, stats, s, bnb, radd, ld_r);
    else *st = launch_stream<K, 2, 1, TRB>(A, lda, Bw, Cmat, ldc, M, N, stats, s, bnb, radd, ld_r);
    return true;
  }
  if constexpr (K <= 64) *st = launch_stream<K, 1, 2, TRB>(A, lda, Bw, Cmat, ldc, M, N, stats, s, bnb, radd, ld_r);
  else *st = launch_stream<K, 1, 1, TRB>(A, lda, Bw, Cmat, ldc, M, N, stats, s, bnb, radd, ld_r);
  return true;
}

template <int BM, int BN, int WM, int WN, int BK>
hd_status launch_cfg(const void* A, int64_t lda, const void* Bw, void* Cmat, int64_t ldc, int M, int N, int K, float* stats,
                     hipStream_t s) {
  const long long tiles = (long long)((M + BM - 1) / BM) * (N / BN);
  long long g = (tiles + 7) / 8 * 8;
  if (g > 512) g = 512;
  const size_t lds = 2u * (BM + BN) * BK * 2u;
  const auto* a = static_cast<const unsigned short*>(A);
  const auto* b = static_cast<const unsigned short*>(Bw);
  auto* c = static_cast<unsigned short*>(Cmat);
  if (stats)
    (void)hd_launch(pointwise_conv_kernel<BM, BN, WM, WN, BK, true>, dim3((unsigned)g), dim3(256), lds, s, a, lda, b, c, ldc, M,
                    N, K, stats);
  else
    (void)hd_launch(pointwise_conv_kernel<BM, BN, WM, WN, BK, false>, dim3((unsigned)g), dim3(256), lds, s, a, lda, b, c, ldc, M,
                    N, K, stats);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

template <int BK>
hd_status launch_bk(const void* A, int64_t lda, const void* Bw, void* Cmat, int64_t ldc, int M, int N, int K, float* stats,
                    hipStream_t s) {
  const int bn = N % 128 == 0 ? 128 : (N % 64 == 0 ? 64 : 32);
  // 128-row tiles unless that leaves CUs idle (the 20x20 / 40x40 levels)
  const bool small = (long long)((M + 127) / 128) * (N / bn) < 256;
  if (bn == 128)
    return small ? launch_cfg<64, 128, 2, 2, BK>(A, lda, Bw, Cmat, ldc, M, N, K, stats, s)
                 : launch_cfg<128, 128, 2, 2, BK>(A, lda, Bw, Cmat, ldc, M, N, K, stats, s);
  if (bn == 64)
    return small ? launch_cfg<64, 64, 2, 2, BK>(A, lda, Bw, Cmat, ldc, M, N, K, stats, s)
                 : launch_cfg<128, 64, 2, 2, BK>(A, lda, Bw, Cmat, ldc, M, N, K, stats, s);
  return launch_cfg<128, 32, 4, 1, BK>(A, lda, Bw, Cmat, ldc, M, N, K, stats, s);
}

// ---- weight gradient: dW[Cout, Cin] = dY[R, Cout]^T . X[R, Cin] -------------------------------------------------
// The reduction runs over the ROWS of both operands, which are the slow index of their channels-last storage, so the
// MFMA fragments (8 consecutive reduction indices per lane) are columns of the staged tiles: rows are staged
// global -> registers -> LDS as they lie (coalesced 16-byte chunks), and each fragment is two ds_read_b64_tr_b16
// (the LDS transpose read: a 16-lane group fetches 4 rows x 16 columns and hands lane i column i).  Rows are padded by
// 64 bytes so that the 8 (group, row) windows of a 32-lane half fall on disjoint banks.
// A workgroup owns a (64 TM) x (64 TN) tile of dW and one of S slices of the rows; two register stages of 64 rows each
// are in flight while the current one is multiplied.  Partial tiles go to a float32 slab [S][Cout][Cin] with plain
// stores (no zero-fill, no atomics) and wgrad_reduce_kernel sums the S slabs into the bf16 gradient — two launches where
// the library's split-K path takes a zero-fill, the convolution and a cast.
// BNB: dY is formed on the fly from d(out) and the BatchNorm input y (hd_bn_fuse.h) while the tile is staged; the
// workgroups of the first row slice and first k-tile also publish dgamma / dbeta of their channels.
// Workgroups with blockIdx.y >= n_splits do not multiply at all: they sum the row-sliced partials of the PREVIOUS weight
// gradient launch (`job`), which rides along instead of costing a launch of its own (5.9 us each, 45 per YOLOv5s step).
template <int TM, int TN, bool BNB>
__global__ __launch_bounds__(256, 2) void pointwise_wgrad_kernel(const unsigned short* __restrict__ dY, long long ld_dy,
                                                              const unsigned short* __restrict__ X, long long ld_x, int M, int N,
                                                              int K, int rows_per_split, float* __restrict__ slab, int n_splits,
                                                              const hdfuse::BnBwd bn, const hdfuse::ReduceJob job) {
  constexpr int BNT = 64 * TM, BKT = 64 * TN, ROWS = 64;
  constexpr int SN = BNT * 2 + 64, SK = BKT * 2 + 64;  // padded row strides (bytes)
  constexpr int CN = BNT / 8, CK = BKT / 8;             // 16-byte chunks per row
  constexpr int IT_N = ROWS * CN / 256, IT_K = ROWS * CK / 256;
  static_assert(256 % CN == 0, "a thread's chunk index must not depend on the load index");
  __shared__ float4 s_part[16][17];
  __shared__ float s_bn[BNB ? 4 * BNT : 1];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [64 rows of dY tile | 64 rows of X tile]
  if ((int)blockIdx.y >= n_splits) {  // the reduction of the previous launch's partials
    const long long blk = (long long)(blockIdx.y - n_splits) * gridDim.x + blockIdx.x;
    if (blk < hdfuse::reduce_blocks(job.count)) hdfuse::reduce_block(job, blk, s_part);
    return;
  }
  unsigned char* tY = smem;
  unsigned char* tX = smem + ROWS * SN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int tiles_k = K / BKT;
  const int n0 = (blockIdx.x / tiles_k) * BNT, k0 = (blockIdx.x % tiles_k) * BKT;
  const int split = blockIdx.y;
  const int m_begin = split * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);
  const int steps = m_end > m_begin ? (m_end - m_begin + ROWS - 1) / ROWS : 0;

  hdfuse::Place pa;
  pa.base = dY + n0 + (tid % CN) * 8;
  pa.ld = ld_dy;
  hdfuse::Coef8 cf;
  if (BNB) {
    pa.set(dY, ld_dy, bn, n0 + (tid % CN) * 8);
    hdfuse::fill_coefs(bn, n0, BNT, s_bn, split == 0 && blockIdx.x % tiles_k == 0);
    __syncthreads();
    cf.load(s_bn, BNT, (tid % CN) * 8);
  }

  u32x4 ry[2][IT_N], rx[2][IT_K], rz[BNB ? 2 : 1][BNB ? IT_N : 1];
  auto issue = [&](int t, u32x4* py, u32x4* px, u32x4* pz) {
    const int mb = m_begin + t * ROWS;
#pragma unroll
    for (int i = 0; i < IT_N; ++i) {
      const int id = i * 256 + tid;
      int m = mb + id / CN;
      m = m < M ? m : M - 1;
      py[i] = *reinterpret_cast<const u32x4*>(pa.base + (long long)m * pa.ld);
      if (BNB) pz[i] = *reinterpret_cast<const u32x4*>(bn.y + (long long)m * bn.ld_y + n0 + (id % CN) * 8);
    }
#pragma unroll
    for (int i = 0; i < IT_K; ++i) {
      const int id = i * 256 + tid;
      int m = mb + id / CK;
      m = m < M ? m : M - 1;
      px[i] = *reinterpret_cast<const u32x4*>(X + (long long)m * ld_x + k0 + (id % CK) * 8);
    }
  };
  auto stash = [&](int t, const u32x4* py, const u32x4* px, const u32x4* pz) {  // rows past the slice end count as zero
    const int mb = m_begin + t * ROWS;
    const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < IT_N; ++i) {
      const int id = i * 256 + tid, row = id / CN;
      const u32x4 v = BNB ? hdfuse::chunk(py[i], pz[i], cf, bn.act) : py[i];
      *reinterpret_cast<u32x4*>(tY + row * SN + (id % CN) * 16) = (mb + row < m_end) ? v : zero;
    }
#pragma unroll
    for (int i = 0; i < IT_K; ++i) {
      const int id = i * 256 + tid, row = id / CK;
      *reinterpret_cast<u32x4*>(tX + row * SK + (id % CK) * 16) = (mb + row < m_end) ? px[i] : zero;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  auto multiply = [&]() {
#pragma unroll
    for (int ks = 0; ks < ROWS / 16; ++ks) {
      bf16x8 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = tr_fragment(tY, SN, (wn * TM + i) * 32, ks * 16, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = tr_fragment(tX, SK, (wk * TN + j) * 32, ks * 16, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };

  u32x4* const rz0 = rz[0];
  u32x4* const rz1 = rz[BNB ? 1 : 0];
  if (steps > 0) issue(0, ry[0], rx[0], rz0);
  if (steps > 1) issue(1, ry[1], rx[1], rz1);
  for (int t = 0; t < steps; t += 2) {
    stash(t, ry[0], rx[0], rz0);
    if (t + 2 < steps) issue(t + 2, ry[0], rx[0], rz0);
    __syncthreads();
    multiply();
    __syncthreads();
    if (t + 1 < steps) {
      stash(t + 1, ry[1], rx[1], rz1);
      if (t + 3 < steps) issue(t + 3, ry[1], rx[1], rz1);
      __syncthreads();
      multiply();
      __syncthreads();
    }
  }

  // C/D map: col (k) = lane & 31, row (n) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float* out = slab + (size_t)split * N * K;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int k = k0 + (wk * TN + j) * 32 + (lane & 31);
      const int nb = n0 + (wn * TM + i) * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) out[(size_t)(nb + (r & 3) + 8 * (r >> 2)) * K + k] = acc[i][j][r];
    }
}

// 32 x 32 tiles: channel counts that are multiples of 32 but not of 64 (the 32-channel layers of YOLOv5s at 160 x 160).  One
// tile is one wave's MFMA, so the four waves split the 64-row stage along the REDUCTION instead (16 rows each) and fold their
// partial tiles in LDS at the end; rows of 64 bytes need no padding for the transpose reads (four rows = four bank quarters).
template <bool BNB>
__global__ __launch_bounds__(256) void pointwise_wgrad32_kernel(const unsigned short* __restrict__ dY, long long ld_dy,
                                                                const unsigned short* __restrict__ X, long long ld_x, int M, int N,
                                                                int K, int rows_per_split, float* __restrict__ slab, int n_splits,
                                                                const hdfuse::BnBwd bn, const hdfuse::ReduceJob job) {
  constexpr int ROWS = 64, SR = 64;  // bytes per LDS row of a 32-channel tile
  __shared__ float4 s_part[16][17];
  __shared__ float s_bn[BNB ? 4 * 32 : 1];
  __shared__ __attribute__((aligned(16))) unsigned char tY[ROWS * SR];
  __shared__ __attribute__((aligned(16))) unsigned char tX[ROWS * SR];
  __shared__ float s_red[4][32][33];
  if ((int)blockIdx.y >= n_splits) {
    const long long blk = (long long)(blockIdx.y - n_splits) * gridDim.x + blockIdx.x;
    if (blk < hdfuse::reduce_blocks(job.count)) hdfuse::reduce_block(job, blk, s_part);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_k = K / 32;
  const int n0 = (blockIdx.x / tiles_k) * 32, k0 = (blockIdx.x % tiles_k) * 32;
  const int split = blockIdx.y;
  const int m_begin = split * rows_per_split;
  const int m_end = min(M, m_begin + rows_per_split);
  const int steps = m_end > m_begin ? (m_end - m_begin + ROWS - 1) / ROWS : 0;
  const int row_t = tid >> 2, ch_t = tid & 3;  // one 16-byte chunk of each operand per thread per stage

  hdfuse::Place pa;
  pa.base = dY + n0 + ch_t * 8;
  pa.ld = ld_dy;
  hdfuse::Coef8 cf;
  if (BNB) {
    pa.set(dY, ld_dy, bn, n0 + ch_t * 8);
    hdfuse::fill_coefs(bn, n0, 32, s_bn, split == 0 && blockIdx.x % tiles_k == 0);
    __syncthreads();
    cf.load(s_bn, 32, ch_t * 8);
  }
  u32x4 ry[2], rx[2], rz[2];
  auto issue = [&](int t, int st) {
    int m = m_begin + t * ROWS + row_t;
    m = m < M ? m : M - 1;
    ry[st] = *reinterpret_cast<const u32x4*>(pa.base + (long long)m * pa.ld);
    rx[st] = *reinterpret_cast<const u32x4*>(X + (long long)m * ld_x + k0 + ch_t * 8);
    if (BNB) rz[st] = *reinterpret_cast<const u32x4*>(bn.y + (long long)m * bn.ld_y + n0 + ch_t * 8);
  };
  auto stash = [&](int t, int st) {  // rows past the slice end count as zero
    const bool in = m_begin + t * ROWS + row_t < m_end;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    const u32x4 v = BNB ? hdfuse::chunk(ry[st], rz[st], cf, bn.act) : ry[st];
    *reinterpret_cast<u32x4*>(tY + row_t * SR + ch_t * 16) = in ? v : zero;
    *reinterpret_cast<u32x4*>(tX + row_t * SR + ch_t * 16) = in ? rx[st] : zero;
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  if (steps > 0) issue(0, 0);
  if (steps > 1) issue(1, 1);
  for (int t = 0; t < steps; ++t) {
    const int st = t & 1;
    if (st == 0) stash(t, 0); else stash(t, 1);
    if (t + 2 < steps) { if (st == 0) issue(t + 2, 0); else issue(t + 2, 1); }
    __syncthreads();
    const bf16x8 a = tr_fragment(tY, SR, 0, 16 * wave, lane);
    const bf16x8 b = tr_fragment(tX, SR, 0, 16 * wave, lane);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    __syncthreads();
  }
  // C/D map: col (k) = lane & 31, row (n) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int r = 0; r < 16; ++r) s_red[wave][(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][lane & 31] = acc[r];
  __syncthreads();
  float* out = slab + (size_t)split * N * K;
  for (int i = tid; i < 32 * 32; i += 256) {
    const int n = i >> 5, k = i & 31;
    out[(size_t)(n0 + n) * K + k0 + k] = (s_red[0][n][k] + s_red[1][n][k]) + (s_red[2][n][k] + s_red[3][n][k]);
  }
}

// Sum over the row slices on its own launch (the last weight gradient of a backward pass, or a caller that does not chain).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const hdfuse::ReduceJob job) {
  __shared__ float4 s_part[16][17];
  hdfuse::reduce_block(job, blockIdx.x, s_part);
}

// ---- 3x3 / stride 1 / pad 1 weight gradient -------------------------------------------------------------------
//     dW[n][ky][kx][k] = sum over (b, h, w) of dY[b, h, w, n] * X[b, h + ky - 1, w + kx - 1, k]
// The same transposed-fragment scheme, one image row at a time: a workgroup owns a 64 x 64 (n, k) tile for ALL nine
// taps (9 x 16 accumulator registers per lane) and a slice of image rows.  LDS holds a ring of three X rows
// (h - 1, h, h + 1), each with one zero pixel of halo on both sides and zero padding up to a multiple of 16 pixels, and
// the dY row; the nine taps are nine (row slot, pixel shift) views of that ring, so borders need no masks — out-of-image
// rows are stored as zeros, out-of-image pixels are the halo.  Per 16 pixels a wave issues one dY fragment, nine X
// fragments and nine MFMAs.  The next X row and dY row travel in registers while the current row is multiplied.
//
// Global loads are buffer instructions: a descriptor per (image row, 64-channel tile) built from uniform values, whose range
// ends with the tile's last channel of the row's last pixel; the lane's byte offset inside a block of 32 pixels in ONE VGPR per
// tensor for the whole kernel, the block's offset (i * 32 pixels) in an SGPR — instead of a 64-bit address per load and lane
// (the 80-pixel instantiation spilled 12 bytes per lane with those; every instantiation is now scratch-free).  Lanes past the
// row's last pixel (rows are not multiples of 32 pixels) fall outside the range: the hardware returns zeros without a memory
// access (range check on voffset + soffset, profiles/r05_soffset_probe.log; row strides are >= the tile's 64 channels), and
// a request for a row that is not this slice's gets an EMPTY range — so every load is unconditional, no row past the slice
// is fetched, and the compiler's vmcnt counts stay exact.
//
// History of the variants measured on this kernel (profiles/r05b_wgrad_rows.md): a 512-thread form whose two halves walked
// two consecutive slices and folded their accumulators through LDS (round 4: half the slab) ran 10 - 35 % slower per launch
// than two independent 256-thread workgroups per CU (its halves share every barrier and so multiply and store in lock step)
// and was removed in round 5; two row sets in flight instead of one changed nothing (the row loop is not bound by the load
// round trip).
template <int W>
__global__ __launch_bounds__(256, W > 80 ? 1 : 2) void conv3x3_wgrad_kernel(
    const unsigned short* __restrict__ dY, long long ld_dy, const unsigned short* __restrict__ X, long long ld_x, int B, int H,
    int N, int K, int rows_per_slice, int images_per_slice, float* __restrict__ slab, int n_slices, const hdfuse::ReduceJob job) {
  __shared__ float4 s_part[16][17];
  constexpr int WP = (W + 15) / 16 * 16, KS = WP / 16;
  constexpr int SX = 64 * 2 + 64;                 // bytes per pixel row of a 64-channel tile, padded
  constexpr int XROW = (WP + 2) * SX, DYROW = WP * SX;
  constexpr int RING = 3 * XROW + DYROW;
  constexpr int CH = W * 8;                       // 16-byte chunks per image row of a 64-channel tile
  constexpr int IT = (CH + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [3 X rows | dY row]
  if ((int)blockIdx.y >= n_slices) {  // the slice sum of the previous weight-gradient launch rides along (hd_wgrad_job)
    const long long blk = (long long)(blockIdx.y - n_slices) * gridDim.x + blockIdx.x;
    if (blk < hdfuse::reduce_blocks(job.count)) hdfuse::reduce_block(job, blk, s_part);
    return;
  }
  unsigned char* xs = smem;
  unsigned char* ys = xs + 3 * XROW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int tiles_k = K / 64;
  const int n0 = (blockIdx.x / tiles_k) * 64, k0 = (blockIdx.x % tiles_k) * 64;
  // a slice is a range of image rows of one image (rows_per_slice < H), or images_per_slice whole images
  const int slice = (int)blockIdx.y;
  const int spi = (H + rows_per_slice - 1) / rows_per_slice;  // slices per image (1 when a slice holds whole images)
  const int b_first = spi > 1 ? slice / spi : slice * images_per_slice;
  const int b_last = spi > 1 ? b_first + 1 : min(B, b_first + images_per_slice);
  const int h0 = spi > 1 ? (slice % spi) * rows_per_slice : 0;
  const int h1 = min(H, h0 + rows_per_slice);

  const u32x4 zero = {0u, 0u, 0u, 0u};
  int b = b_first;
  const int vo_x = ((tid >> 3) * (int)ld_x + (tid & 7) * 8) * 2, vo_y = ((tid >> 3) * (int)ld_dy + (tid & 7) * 8) * 2;
  const int sx32 = 32 * (int)ld_x * 2, sy32 = 32 * (int)ld_dy * 2;
  const int rec_x = ((W - 1) * (int)ld_x + 64) * 2, rec_y = ((W - 1) * (int)ld_dy + 64) * 2;
  auto load_x = [&](int h, u32x4* r, bool on) {  // image row h of X (zeros outside the image); !on: empty range, no access
    const int hc = min(max(h, 0), H - 1);
    const unsigned short* base = X + ((long long)(b * H + hc) * W) * ld_x + k0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, on ? rec_x : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < IT; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo_x, i * sx32, 0);
  };
  auto load_y = [&](int h, u32x4* r, bool on) {
    const int hc = min(h, H - 1);
    const unsigned short* base = dY + ((long long)(b * H + hc) * W) * ld_dy + n0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, on ? rec_y : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < IT; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo_y, i * sy32, 0);
  };
  auto stash_x = [&](int h, const u32x4* r) {
    unsigned char* slot = xs + ((h + 3) % 3) * XROW;
    const bool in = h >= 0 && h < H;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int id = i * 256 + tid;
      if (CH % 256 == 0 || id < CH) *reinterpret_cast<u32x4*>(slot + ((id >> 3) + 1) * SX + (id & 7) * 16) = in ? r[i] : zero;
    }
  };
  auto stash_y = [&](const u32x4* r) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int id = i * 256 + tid;
      if (CH % 256 == 0 || id < CH) *reinterpret_cast<u32x4*>(ys + (id >> 3) * SX + (id & 7) * 16) = r[i];
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  // Start of an image's rows: the four rows the first multiply needs (X rows h0 - 1, h0, h0 + 1 and dY row h0) are requested
  // back to back — ONE memory round trip, not four load -> wait -> store rounds — and the first image's requests are issued
  // before the ring is zeroed, so the 75 KB of LDS stores run under that round trip too.  Every row register has ONE defining
  // load site per loop level: a second copy of these loads outside the image loop made the compiler carry the rows through
  // register copies and wait for ALL outstanding loads at the top of every row's multiply (seen in the ISA, never shipped).
  u32x4 rx[IT], ry[IT];
  for (b = b_first; b < b_last; ++b) {
    if (b > b_first) __syncthreads();  // the previous image's last row has been read
    u32x4 t0[IT], t1[IT];
    load_x(h0 - 1, t0, true); load_x(h0, t1, true); load_x(h0 + 1, rx, true); load_y(h0, ry, true);
    if (b == b_first) {
      for (int i = tid; i < RING / 16; i += 256) reinterpret_cast<uint4*>(xs)[i] = make_uint4(0, 0, 0, 0);
      __syncthreads();
    }
    stash_x(h0 - 1, t0); stash_x(h0, t1); stash_x(h0 + 1, rx); stash_y(ry);
    load_x(h0 + 2, rx, h0 + 1 < h1);
    load_y(h0 + 1, ry, h0 + 1 < h1);
    for (int h = h0; h < h1; ++h) {
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 a = tr_fragment(ys, SX, wn * 32, ks * 16, lane);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const unsigned char* slot = xs + ((h + ky - 1 + 3) % 3) * XROW;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const bf16x8 bq = tr_fragment(slot, SX, wk * 32, ks * 16 + kx, lane);
            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, acc[ky * 3 + kx], 0, 0, 0);
          }
        }
      }
      __syncthreads();
      if (h + 1 < h1) {
        stash_x(h + 2, rx);  // into the slot row h - 1 just left
        stash_y(ry);
      }
      load_x(h + 3, rx, h + 2 < h1);  // requested on every path: the rows of output row h + 2, an empty range past the slice
      load_y(h + 2, ry, h + 2 < h1);
    }
  }

  // C/D map: col (k) = lane & 31, row (n) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); slab[slice][n][tap][k]
  float* out = slab + (size_t)blockIdx.y * N * 9 * K;
  const int k = k0 + wk * 32 + (lane & 31);
  const int nb = n0 + wn * 32 + 4 * (lane >> 5);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[((size_t)(nb + (r & 3) + 8 * (r >> 2)) * 9 + t) * K + k] = acc[t][r];
}

// Stride 2 (the downsampling 3x3 convolutions): output pixel (ho, wo) reads input (2 ho + ky - 1, 2 wo + kx - 1).  An input
// row is stored de-interleaved — even pixels E[j] = x[2 j] and odd pixels O[j] = x[2 j + 1], each with the one-pixel zero
// halo — so the three column taps are again contiguous views: kx = 0 -> O[wo - 1], kx = 1 -> E[wo], kx = 2 -> O[wo].  The
// ring holds input rows 2 ho - 1, 2 ho, 2 ho + 1; advancing one output row brings in two input rows.  H, W: OUTPUT size.
template <int W>
__global__ __launch_bounds__(256, W > 40 ? 1 : 2) void conv3x3s2_wgrad_kernel(
    const unsigned short* __restrict__ dY, long long ld_dy, const unsigned short* __restrict__ X, long long ld_x, int B, int H,
    int N, int K, int rows_per_slice, int images_per_slice, float* __restrict__ slab, int n_slices, const hdfuse::ReduceJob job) {
  __shared__ float4 s_part[16][17];
  constexpr int WP = (W + 15) / 16 * 16, KS = WP / 16;
  constexpr int SX = 64 * 2 + 64;
  constexpr int PLANE = (WP + 2) * SX, XROW = 2 * PLANE, DYROW = WP * SX;
  constexpr int RING = 3 * XROW + DYROW;
  constexpr int CHX = 2 * W * 8, CHY = W * 8;  // 16-byte chunks per input row / per dY row of a 64-channel tile
  constexpr int ITX = (CHX + 255) / 256, ITY = (CHY + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [3 input rows (E | O planes) | dY row]
  if ((int)blockIdx.y >= n_slices) {  // the slice sum of the previous weight-gradient launch rides along (hd_wgrad_job)
    const long long blk = (long long)(blockIdx.y - n_slices) * gridDim.x + blockIdx.x;
    if (blk < hdfuse::reduce_blocks(job.count)) hdfuse::reduce_block(job, blk, s_part);
    return;
  }
  unsigned char* xs = smem;
  unsigned char* ys = xs + 3 * XROW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int tiles_k = K / 64;
  const int n0 = (blockIdx.x / tiles_k) * 64, k0 = (blockIdx.x % tiles_k) * 64;
  const int Hi = 2 * H, Wi = 2 * W;
  const int slice = (int)blockIdx.y;
  const int spi = (H + rows_per_slice - 1) / rows_per_slice;
  const int b_first = spi > 1 ? slice / spi : slice * images_per_slice;
  const int b_last = spi > 1 ? b_first + 1 : min(B, b_first + images_per_slice);
  const int h0 = spi > 1 ? (slice % spi) * rows_per_slice : 0;
  const int h1 = min(H, h0 + rows_per_slice);

  const u32x4 zero = {0u, 0u, 0u, 0u};
  int b = b_first;
  // buffer row loads as in conv3x3_wgrad_kernel: one descriptor per (row, 64-channel tile), one offset VGPR per tensor
  const int vo_x = ((tid >> 3) * (int)ld_x + (tid & 7) * 8) * 2, vo_y = ((tid >> 3) * (int)ld_dy + (tid & 7) * 8) * 2;
  const int sx32 = 32 * (int)ld_x * 2, sy32 = 32 * (int)ld_dy * 2;
  const int rec_x = ((Wi - 1) * (int)ld_x + 64) * 2, rec_y = ((W - 1) * (int)ld_dy + 64) * 2;
  auto load_x = [&](int r, u32x4* v, bool on) {  // input row r (zeros outside the image); !on: empty range, no memory access
    const int rc = min(max(r, 0), Hi - 1);
    const unsigned short* base = X + ((long long)(b * Hi + rc) * Wi) * ld_x + k0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, on ? rec_x : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < ITX; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo_x, i * sx32, 0);
  };
  auto load_y = [&](int h, u32x4* v, bool on) {
    const int hc = min(h, H - 1);
    const unsigned short* base = dY + ((long long)(b * H + hc) * W) * ld_dy + n0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, on ? rec_y : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < ITY; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo_y, i * sy32, 0);
  };
  auto stash_x = [&](int r, const u32x4* v) {
    unsigned char* slot = xs + ((r + 3) % 3) * XROW;
    const bool in = r >= 0 && r < Hi;
#pragma unroll
    for (int i = 0; i < ITX; ++i) {
      const int id = i * 256 + tid;
      if (CHX % 256 == 0 || id < CHX) {
        const int p = id >> 3;
        *reinterpret_cast<u32x4*>(slot + (p & 1) * PLANE + ((p >> 1) + 1) * SX + (id & 7) * 16) = in ? v[i] : zero;
      }
    }
  };
  auto stash_y = [&](const u32x4* v) {
#pragma unroll
    for (int i = 0; i < ITY; ++i) {
      const int id = i * 256 + tid;
      if (CHY % 256 == 0 || id < CHY) *reinterpret_cast<u32x4*>(ys + (id >> 3) * SX + (id & 7) * 16) = v[i];
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  // start of an image's rows as in conv3x3_wgrad_kernel: the first four rows in one round trip, the first image's under the
  // ring's zeroing
  u32x4 ra[ITX], rb[ITX], ry[ITY];
  for (b = b_first; b < b_last; ++b) {
    if (b > b_first) __syncthreads();
    u32x4 t0[ITX];
    load_x(2 * h0 - 1, t0, true); load_x(2 * h0, ra, true); load_x(2 * h0 + 1, rb, true); load_y(h0, ry, true);
    if (b == b_first) {
      for (int i = tid; i < RING / 16; i += 256) reinterpret_cast<uint4*>(xs)[i] = make_uint4(0, 0, 0, 0);
      __syncthreads();
    }
    stash_x(2 * h0 - 1, t0); stash_x(2 * h0, ra); stash_x(2 * h0 + 1, rb); stash_y(ry);
    load_x(2 * h0 + 2, ra, h0 + 1 < h1);
    load_x(2 * h0 + 3, rb, h0 + 1 < h1);
    load_y(h0 + 1, ry, h0 + 1 < h1);
    for (int h = h0; h < h1; ++h) {
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 a = tr_fragment(ys, SX, wn * 32, ks * 16, lane);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const unsigned char* slot = xs + ((2 * h + ky - 1 + 3) % 3) * XROW;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            // kx = 0: O[wo - 1] (pixel row wo), kx = 1: E[wo] (pixel row wo + 1), kx = 2: O[wo] (pixel row wo + 1)
            const bf16x8 bq = tr_fragment(slot + (kx == 1 ? 0 : PLANE), SX, wk * 32, ks * 16 + (kx == 0 ? 0 : 1), lane);
            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, acc[ky * 3 + kx], 0, 0, 0);
          }
        }
      }
      __syncthreads();
      if (h + 1 < h1) {
        stash_x(2 * h + 2, ra);  // into the slots input rows 2 h - 1 and 2 h just left
        stash_x(2 * h + 3, rb);
        stash_y(ry);
      }
      load_x(2 * h + 4, ra, h + 2 < h1);  // on every path; an empty range past the slice
      load_x(2 * h + 5, rb, h + 2 < h1);
      load_y(h + 2, ry, h + 2 < h1);
    }
  }

  float* out = slab + (size_t)blockIdx.y * N * 9 * K;
  const int k = k0 + wk * 32 + (lane & 31);
  const int nb = n0 + wn * 32 + 4 * (lane >> 5);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[((size_t)(nb + (r & 3) + 8 * (r >> 2)) * 9 + t) * K + k] = acc[t][r];
}

// Row widths with an instantiation: the 640^2 / 1280^2 pyramids (20, 40, 80) and the 832^2 ones (26, 52, 104).  One image
// row (+ halo) must fit the LDS ring, which ends at 104 pixels for the stride-2 weight gradient and the forward kernel.
static inline bool conv3x3_width_ok(int W) { return W == 20 || W == 26 || W == 40 || W == 52 || W == 80 || W == 104; }

struct Wgrad3Plan {
  int tiles, rows_per_slice, images_per_slice, slices;
  int partials;  // slab partials = grid rows that multiply (= slices)
};
inline Wgrad3Plan wgrad3_plan(int B, int H, int Cin, int Cout, int W = 0, int stride = 1) {
  // Every workgroup writes a 64 x 9 x 64 float32 partial (147 KB) that the reduce kernel reads back, so the slab traffic
  // grows with the workgroup count while the operands do not (75 MB for the 128-channel 40 x 40 layer at 512 workgroups
  // against 13 MB of operands): the partial kernel is bound by that write.  Measured per launch in the step, W = 40 / 80 / 20
  // stride 1: 512 workgroups 24.5 / 24.1 / 26.5 us (+ 6.9 us reduce); 256: 23.0 / 22.8 / 29.8 (+ 5.9); 128: 36 / 33 / 47
  // (+ 4.9, the row loop then dominates).  Two rows of loads in flight instead of one changed nothing.  Round 5 (final
  // kernels, per-launch probe incl. the riding slice sum, profiles/r05b_wgrad_rows.md): 256 slices x tiles stays the
  // choice — 512 and 1024 are slower on every YOLOv5s shape but the 64-channel 80-pixel one (20.0 - 21.0 against 18.8 us).
  Wgrad3Plan p;
  p.tiles = (Cout / 64) * (Cin / 64);
  int want = 256 / p.tiles;
  if (want < 1) want = 1;
  if (want >= B) {  // one image, a range of rows
    int spi = want / B;
    int rps = (H + spi - 1) / spi;
    if (rps < 4 && H >= 4) rps = 4;  // two halo rows per slice: keep them under a third of the loads
    p.rows_per_slice = rps;
    p.images_per_slice = 1;
    p.slices = B * ((H + rps - 1) / rps);
  } else {          // whole images
    p.rows_per_slice = H;
    p.images_per_slice = (B + want - 1) / want;
    p.slices = (B + p.images_per_slice - 1) / p.images_per_slice;
  }
  p.partials = p.slices;
  return p;
}

struct WgradPlan {
  int tm, tn, tiles, splits, rows_per_split;
};
inline WgradPlan wgrad_plan(int64_t R, int Cin, int Cout) {
  WgradPlan p;
  if (Cin % 64 || Cout % 64) {  // 32 x 32 tiles (pointwise_wgrad32_kernel): tm = tn = 0 marks them
    p.tm = p.tn = 0;
    p.tiles = (Cout / 32) * (Cin / 32);
    int s32 = 512 / p.tiles;
    const int64_t max_s32 = (R + 255) / 256;  // at least four 64-row steps per slice
    if (s32 > max_s32) s32 = (int)max_s32;
    if (s32 < 1) s32 = 1;
    int64_t rps32 = (R + s32 - 1) / s32;
    rps32 = (rps32 + 63) / 64 * 64;
    p.rows_per_split = (int)rps32;
    p.splits = (int)((R + rps32 - 1) / rps32);
    return p;
  }
  // 128-wide tiles read each operand once per tile row / column; for a 128 x 128 gradient that is ONE tile, whose
  // ~200 row slices would write (and the reduce kernel read) as many slab bytes as the operands themselves: four
  // 64 x 64 tiles with a quarter of the slices instead
  const bool big = (int64_t)Cin * Cout > 128 * 128;
  p.tm = Cout % 128 == 0 && big ? 2 : 1;
  p.tn = Cin % 128 == 0 && big ? 2 : 1;
  p.tiles = (Cout / (64 * p.tm)) * (Cin / (64 * p.tn));
  int s = 512 / p.tiles;  // ~2 workgroups per CU (256 measured equal within noise)
  const int64_t max_s = (R + 127) / 128;  // at least two 64-row steps per slice
  if (s > max_s) s = (int)max_s;
  if (s < 1) s = 1;
  int64_t rps = (R + s - 1) / s;
  rps = (rps + 63) / 64 * 64;
  p.rows_per_split = (int)rps;
  p.splits = (int)((R + rps - 1) / rps);
  return p;
}

}  // namespace

// pointwise_longk.hip: long reductions on few rows (Cin = 512 / 1024, the 20x20 / 40x40 levels)
int hd_pointwise_longk_supported(int64_t R, int Cin, int Cout);
hd_status hd_pointwise_longk(const void* x, int64_t ldx, const void* w, int64_t R, int Cin, int Cout, void* y, int64_t ldy,
                             float* stats, hipStream_t s);

extern "C" int hd_pointwise_conv_supported(int64_t R, int Cin, int Cout) {
  return R >= 1 && R <= 0x7fffffff && Cin >= 32 && Cin % 32 == 0 && Cout >= 32 && Cout % 32 == 0 && Cin <= 8192 &&
         Cout <= 1024;  // <= 8 column tiles: the persistent grid (a multiple of 8) keeps one column tile per workgroup
}

extern "C" hd_status hd_pointwise_conv(const void* x, int64_t x_row_stride, const void* w, int64_t R, int Cin, int Cout,
                                       void* y, int64_t y_row_stride, float* stats, hd_stream stream) {
  if (!x || !w || !y) return HD_ERR_NULL;
  if (!hd_pointwise_conv_supported(R, Cin, Cout)) return HD_ERR_SHAPE;
  if (x_row_stride == 0) x_row_stride = Cin;
  if (y_row_stride == 0) y_row_stride = Cout;
  if (x_row_stride < Cin || y_row_stride < Cout || x_row_stride % 8 || y_row_stride % 8) return HD_ERR_SHAPE;
  if (!hd_aligned16(x) || !hd_aligned16(w) || !hd_aligned16(y)) return HD_ERR_ALIGN;
  if (hd_bn_act_replicas() != kStatReplicas) return HD_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  hd_status st = HD_OK;
  const int M = (int)R;
  if (Cin == 32 && try_stream<32>(x, x_row_stride, w, y, y_row_stride, M, Cout, stats, s, &st)) return st;
  if (Cin == 64 && try_stream<64>(x, x_row_stride, w, y, y_row_stride, M, Cout, stats, s, &st)) return st;
  if (Cin == 128 && try_stream<128>(x, x_row_stride, w, y, y_row_stride, M, Cout, stats, s, &st)) return st;
  if (Cin == 256 && try_stream<256>(x, x_row_stride, w, y, y_row_stride, M, Cout, stats, s, &st)) return st;
  if (hd_pointwise_longk_supported(R, Cin, Cout))
    return hd_pointwise_longk(x, x_row_stride, w, R, Cin, Cout, y, y_row_stride, stats, s);
  if (Cin % 64 == 0) return launch_bk<64>(x, x_row_stride, w, y, y_row_stride, (int)R, Cout, Cin, stats, s);
  return launch_bk<32>(x, x_row_stride, w, y, y_row_stride, (int)R, Cout, Cin, stats, s);
}

// ---- the Detect convolution forward: y[R, pitch] (float32) = x[R, Cin] . w[Cout, Cin]^T + bias ------------------------------
// Cout = na (5 + nc) is no tile multiple (255 for 80 classes): the column range is padded to hd_head_conv_pitch(Cout) and the
// padding columns are written as zeros, so that the rows stay 16-byte aligned and the loss kernels stream them (yolo.hip).
template <int K, int TN, int TM>
static hd_status launch_head(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias, void* y, int64_t ldy, int M,
                             int Np, int n_real, bool accumulate, hipStream_t s) {
  constexpr int BN = TN * 32;
  const int NT = Np / BN;
  const long long RT = ((long long)M + 32 * TM - 1) / (32 * TM);
  using SL = StreamLds<K, TN, TM, false>;
  constexpr int head_patch = 32 * (BN * 4 + 32);  // as in the kernel
  const size_t lds = SL::SLAB_BYTES + 4 * (size_t)(head_patch > SL::WAVE_BYTES ? head_patch : SL::WAVE_BYTES);
  const long long cap = lds > 80 * 1024 ? 256 : 512;
  long long g = (RT + 3) / 4 * NT;
  long long L = 8;
  while (L % NT) L += 8;
  g = (g + L - 1) / L * L;
  if (g > cap) g = cap / L * L;
  const auto* a = static_cast<const unsigned short*>(x);
  const auto* b = static_cast<const unsigned short*>(w);
  auto* c = static_cast<unsigned short*>(y);
  const hdfuse::BnBwd none = {};
  const unsigned short* flag = accumulate ? c : nullptr;  // only tested against null
  const bool deep = K >= 128 && RT * NT >= kDeepBlocks * g * 4;
  if (deep)
    (void)hd_launch(pointwise_stream_kernel<K, TN, TM, false, false, false, false, (K >= 128), true>, dim3((unsigned)g), dim3(256),
                    lds, s, a, (long long)ldx, b, c, (long long)ldy, M, Np, const_cast<float*>(bias), none, flag, (long long)ldw,
                    n_real);
  else
    (void)hd_launch(pointwise_stream_kernel<K, TN, TM, false, false, false, false, false, true>, dim3((unsigned)g), dim3(256), lds,
                    s, a, (long long)ldx, b, c, (long long)ldy, M, Np, const_cast<float*>(bias), none, flag, (long long)ldw,
                    n_real);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" int hd_head_conv_fwd_supported(int64_t R, int Cin, int Cout) {
  const int p = Cout >= 1 ? hd_head_conv_pitch(Cout) : 0;
  return p && R >= 1 && R <= 0x7fffffff && (Cin == 32 || Cin == 64 || Cin == 128 || (Cin % 256 == 0 && Cin >= 256 && Cin <= 2048));
}

extern "C" hd_status hd_head_conv_fwd(const void* x, int64_t x_row_stride, const void* w, const float* bias, int64_t R, int Cin,
                                      int Cout, float* y, int64_t y_row_stride, hd_stream stream) {
  if (!x || !w || !bias || !y) return HD_ERR_NULL;
  if (!hd_head_conv_fwd_supported(R, Cin, Cout)) return HD_ERR_SHAPE;
  const int Np = hd_head_conv_pitch(Cout);
  if (x_row_stride == 0) x_row_stride = Cin;
  if (y_row_stride == 0) y_row_stride = Np;
  if (x_row_stride < Cin || y_row_stride < Np || x_row_stride % 8 || y_row_stride % 4) return HD_ERR_SHAPE;
  if (!hd_aligned16(x) || !hd_aligned16(w) || !hd_aligned16(y)) return HD_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const int M = (int)R;
  // column tile: 64 (float32 patch rows of 256 + 32 bytes: 9 KB per wave), 32 for the narrowest head
  const int bn = Np % 64 == 0 ? 64 : 32;
#define HD_HEAD(KK, XX, WW, BIAS, ACCUM)                                                                                          \
  (bn == 64 ? launch_head<KK, 2, (KK <= 64 ? 2 : 1)>(XX, x_row_stride, WW, Cin, BIAS, y, y_row_stride, M, Np, Cout, ACCUM, s)      \
            : launch_head<KK, 1, (KK <= 64 ? 2 : 1)>(XX, x_row_stride, WW, Cin, BIAS, y, y_row_stride, M, Np, Cout, ACCUM, s))
  if (Cin == 32) return HD_HEAD(32, x, w, bias, false);
  if (Cin == 64) return HD_HEAD(64, x, w, bias, false);
  if (Cin == 128) return HD_HEAD(128, x, w, bias, false);
  // Cin = 256 k: k slices of the reduction, each with its 256 columns of the weight resident; slice 0 adds the bias, the others
  // add their product to y (float32: the partial sums are not rounded in between)
  const auto* xs = static_cast<const unsigned short*>(x);
  const auto* ws = static_cast<const unsigned short*>(w);
  for (int k0 = 0; k0 < Cin; k0 += 256) {
    const hd_status st = HD_HEAD(256, xs + k0, ws + k0, (k0 == 0 ? bias : (const float*)nullptr), (k0 != 0));
    if (st != HD_OK) return st;
  }
  return HD_OK;
#undef HD_HEAD
}

extern "C" int hd_pointwise_wgrad_supported(int64_t R, int Cin, int Cout) {
  // multiples of 64 take the 64 / 128-wide tiles; multiples of 32 (up to 256 channels) the 32 x 32 tiles
  if (!(R >= 1 && R <= 0x7fffffff && Cin >= 32 && Cin % 32 == 0 && Cout >= 32 && Cout % 32 == 0 && Cin <= 8192 && Cout <= 8192))
    return 0;
  return (Cin % 64 == 0 && Cout % 64 == 0) || (Cin <= 256 && Cout <= 256);
}

extern "C" size_t hd_pointwise_wgrad_workspace_bytes(int64_t R, int Cin, int Cout) {
  if (!hd_pointwise_wgrad_supported(R, Cin, Cout)) return 0;
  const WgradPlan p = wgrad_plan(R, Cin, Cout);
  return (size_t)p.splits * Cin * Cout * sizeof(float);
}

static hd_status make_job_view(const hd_wgrad_job* j, hdfuse::ReduceJob* out) {
  *out = hdfuse::ReduceJob{};
  if (!j) return HD_OK;
  if (!j->slab || !j->dw) return HD_ERR_NULL;
  if (j->slices < 1 || j->count < 4 || j->count % 4) return HD_ERR_SHAPE;
  if (!hd_aligned16(j->slab) || (reinterpret_cast<uintptr_t>(j->dw) & 7u)) return HD_ERR_ALIGN;
  out->slab = j->slab;
  out->slices = j->slices;
  out->count = j->count;
  out->dw = static_cast<unsigned short*>(j->dw);
  return HD_OK;
}

// extra grid rows that hold the reduction blocks of a job riding along (0 without one)
static unsigned piggyback_rows(const hdfuse::ReduceJob& job, unsigned grid_x) {
  if (!job.slab) return 0;
  const long long blocks = hdfuse::reduce_blocks(job.count);
  return (unsigned)((blocks + grid_x - 1) / grid_x);
}

extern "C" hd_status hd_wgrad_reduce(const hd_wgrad_job* job, hd_stream stream) {
  if (!job) return HD_ERR_NULL;
  hdfuse::ReduceJob j;
  const hd_status st = make_job_view(job, &j);
  if (st != HD_OK) return st;
  (void)hd_launch(wgrad_reduce_kernel, dim3((unsigned)hdfuse::reduce_blocks(j.count)), dim3(256), 0, (hipStream_t)stream, j);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_pointwise_wgrad_partial(const void* dy, int64_t dy_row_stride, const void* x, int64_t x_row_stride,
                                                int64_t R, int Cin, int Cout, void* dw, void* workspace, size_t workspace_bytes,
                                                const hd_wgrad_job* piggyback, hd_wgrad_job* job_out, hd_stream stream) {
  // (the BNB = true instantiations of the kernels — BatchNorm backward folded into the operand load, round 3: bit-identical and
  // slower — are no longer built: tools/experiments/r05_removed_paths.patch has the entry points that used them)
  if (!dy || !x || !dw || !workspace || !job_out) return HD_ERR_NULL;
  if (!hd_pointwise_wgrad_supported(R, Cin, Cout)) return HD_ERR_SHAPE;
  if (dy_row_stride == 0) dy_row_stride = Cout;
  if (x_row_stride == 0) x_row_stride = Cin;
  if (dy_row_stride % 8 || x_row_stride < Cin || x_row_stride % 8 || dy_row_stride < Cout) return HD_ERR_SHAPE;
  if (!hd_aligned16(dy) || !hd_aligned16(x) || !hd_aligned16(dw) || !hd_aligned16(workspace)) return HD_ERR_ALIGN;
  const WgradPlan p = wgrad_plan(R, Cin, Cout);
  if (workspace_bytes < (size_t)p.splits * Cin * Cout * sizeof(float)) return HD_ERR_WORKSPACE;
  const hdfuse::BnBwd bv{};
  hdfuse::ReduceJob jv;
  const hd_status st = make_job_view(piggyback, &jv);
  if (st != HD_OK) return st;
  hipStream_t s = (hipStream_t)stream;
  const auto* a = static_cast<const unsigned short*>(dy);
  const auto* b = static_cast<const unsigned short*>(x);
  float* slab = static_cast<float*>(workspace);
  const dim3 grid((unsigned)p.tiles, (unsigned)p.splits + piggyback_rows(jv, (unsigned)p.tiles));
  const size_t lds = 64u * ((64u * p.tm * 2u + 64u) + (64u * p.tn * 2u + 64u));
  const int M = (int)R;
#define HD_WG_LAUNCH(TMM, TNN)                                                                                             \
  (void)hd_launch(pointwise_wgrad_kernel<TMM, TNN, false>, grid, dim3(256), lds, s, a, dy_row_stride, b, x_row_stride, M, Cout, \
                  Cin, p.rows_per_split, slab, p.splits, bv, jv)
  if (p.tm == 0) {
    const dim3 g32((unsigned)p.tiles, (unsigned)p.splits + piggyback_rows(jv, (unsigned)p.tiles));
    (void)hd_launch(pointwise_wgrad32_kernel<false>, g32, dim3(256), 0, s, a, dy_row_stride, b, x_row_stride, M, Cout, Cin,
                    p.rows_per_split, slab, p.splits, bv, jv);
  } else if (p.tm == 2 && p.tn == 2) HD_WG_LAUNCH(2, 2);
  else if (p.tm == 2) HD_WG_LAUNCH(2, 1);
  else if (p.tn == 2) HD_WG_LAUNCH(1, 2);
  else HD_WG_LAUNCH(1, 1);
#undef HD_WG_LAUNCH
  HD_RETURN_IF_LAUNCH_FAILED();
  job_out->slab = slab;
  job_out->slices = p.splits;
  job_out->count = (int64_t)Cin * Cout;
  job_out->dw = dw;
  return HD_OK;
}

extern "C" hd_status hd_pointwise_wgrad(const void* dy, int64_t dy_row_stride, const void* x, int64_t x_row_stride, int64_t R,
                                        int Cin, int Cout, void* dw, void* workspace, size_t workspace_bytes,
                                        hd_stream stream) {
  hd_wgrad_job job;
  const hd_status st = hd_pointwise_wgrad_partial(dy, dy_row_stride, x, x_row_stride, R, Cin, Cout, dw, workspace, workspace_bytes,
                                                  nullptr, &job, stream);
  return st != HD_OK ? st : hd_wgrad_reduce(&job, stream);
}

extern "C" int hd_pointwise_conv_resident(int64_t R, int Cin, int Cout) {
  // 1 when hd_pointwise_conv takes a form with the weights resident in LDS (the streaming kernel, or pointwise_longk.hip for
  // Cin = 512 / 1024 on few rows), the ones worth preferring to the library
  if (hd_pointwise_longk_supported(R, Cin, Cout) && hd_pointwise_conv_supported(R, Cin, Cout)) return 1;
  if (!(Cin == 32 || Cin == 64 || Cin == 128 || Cin == 256) || !hd_pointwise_conv_supported(R, Cin, Cout)) return 0;
  const int max_bn = Cin <= 128 ? 128 : 64;
  const int bn = (Cout % 128 == 0 && max_bn >= 128) ? 128 : (Cout % 64 == 0 ? 64 : 32);
  return Cout / bn <= kMaxColumnTiles;
}

extern "C" int hd_pointwise_conv_dgrad_supported(int64_t R, int Cin, int Cout) {
  if (!(Cout == 32 || Cout == 64 || Cout == 128 || Cout == 256) || !hd_pointwise_conv_supported(R, Cout, Cin)) return 0;
  const int max_bn = Cout <= 128 ? 128 : 64;  // the column tile try_stream<Cout, true> picks over the Cin output channels
  const int bn = (Cin % 128 == 0 && max_bn >= 128) ? 128 : (Cin % 64 == 0 ? 64 : 32);
  return Cin / bn <= kMaxColumnTiles;
}

extern "C" hd_status hd_pointwise_conv_dgrad(const void* dy, int64_t dy_row_stride, const void* w, int64_t R, int Cin, int Cout,
                                             void* dx, int64_t dx_row_stride, hd_stream stream) {
  if (!dy || !w || !dx) return HD_ERR_NULL;
  if (!hd_pointwise_conv_dgrad_supported(R, Cin, Cout)) return HD_ERR_SHAPE;  // the reduction runs over Cout
  if (dy_row_stride == 0) dy_row_stride = Cout;
  if (dx_row_stride == 0) dx_row_stride = Cin;
  if (dy_row_stride < Cout || dx_row_stride < Cin || dy_row_stride % 8 || dx_row_stride % 8) return HD_ERR_SHAPE;
  if (!hd_aligned16(dy) || !hd_aligned16(w) || !hd_aligned16(dx)) return HD_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  hd_status st = HD_OK;
  const int M = (int)R;
  if (Cout == 32 && try_stream<32, true>(dy, dy_row_stride, w, dx, dx_row_stride, M, Cin, nullptr, s, &st)) return st;
  if (Cout == 64 && try_stream<64, true>(dy, dy_row_stride, w, dx, dx_row_stride, M, Cin, nullptr, s, &st)) return st;
  if (Cout == 128 && try_stream<128, true>(dy, dy_row_stride, w, dx, dx_row_stride, M, Cin, nullptr, s, &st)) return st;
  if (Cout == 256 && try_stream<256, true>(dy, dy_row_stride, w, dx, dx_row_stride, M, Cin, nullptr, s, &st)) return st;
  return HD_ERR_SHAPE;
}

extern "C" hd_status hd_pointwise_conv_dgrad_add(const void* dy, int64_t dy_row_stride, const void* w, int64_t R, int Cin,
                                                 int Cout, const void* add, int64_t add_row_stride, void* dx,
                                                 int64_t dx_row_stride, hd_stream stream) {
  if (!dy || !w || !dx || !add) return HD_ERR_NULL;
  if (!hd_pointwise_conv_dgrad_supported(R, Cin, Cout)) return HD_ERR_SHAPE;
  if (dy_row_stride == 0) dy_row_stride = Cout;
  if (dx_row_stride == 0) dx_row_stride = Cin;
  if (add_row_stride == 0) add_row_stride = Cin;
  if (dy_row_stride < Cout || dx_row_stride < Cin || add_row_stride < Cin || dy_row_stride % 8 || dx_row_stride % 8 ||
      add_row_stride % 8)
    return HD_ERR_SHAPE;
  if (!hd_aligned16(dy) || !hd_aligned16(w) || !hd_aligned16(dx) || !hd_aligned16(add)) return HD_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  hd_status st = HD_OK;
  const int M = (int)R;
  if (Cout == 32 && try_stream<32, true>(dy, dy_row_stride, w, dx, dx_row_stride, M, Cin, nullptr, s, &st, nullptr, add, add_row_stride)) return st;
  if (Cout == 64 && try_stream<64, true>(dy, dy_row_stride, w, dx, dx_row_stride, M, Cin, nullptr, s, &st, nullptr, add, add_row_stride)) return st;
  if (Cout == 128 && try_stream<128, true>(dy, dy_row_stride, w, dx, dx_row_stride, M, Cin, nullptr, s, &st, nullptr, add, add_row_stride)) return st;
  if (Cout == 256 && try_stream<256, true>(dy, dy_row_stride, w, dx, dx_row_stride, M, Cin, nullptr, s, &st, nullptr, add, add_row_stride)) return st;
  return HD_ERR_SHAPE;
}

int hd_conv3x3_wgrad_supported_v1(int B, int H, int W, int Cin, int Cout, int stride) {
  return (stride == 1 || stride == 2) && B >= 1 && H >= 1 && conv3x3_width_ok(W) && Cin >= 64 && Cin % 64 == 0 &&
         Cout >= 64 && Cout % 64 == 0 && Cin <= 1024 && Cout <= 1024 && (long long)B * H * W * stride * stride <= 0x7fffffff;
}

size_t hd_conv3x3_wgrad_workspace_bytes_v1(int B, int H, int W, int Cin, int Cout, int stride) {
  if (!hd_conv3x3_wgrad_supported_v1(B, H, W, Cin, Cout, stride)) return 0;
  return (size_t)wgrad3_plan(B, H, Cin, Cout, W, stride).partials * Cout * 9 * Cin * sizeof(float);
}

hd_status hd_conv3x3_wgrad_partial_v1(const void* dy, int64_t dy_row_stride, const void* x, int64_t x_row_stride, int B,
                                      int H, int W, int Cin, int Cout, int stride, void* dw, void* workspace,
                                      size_t workspace_bytes, const hd_wgrad_job* piggyback, hd_wgrad_job* job_out,
                                      hd_stream stream) {
  if (!dy || !x || !dw || !workspace || !job_out) return HD_ERR_NULL;
  if (!hd_conv3x3_wgrad_supported_v1(B, H, W, Cin, Cout, stride)) return HD_ERR_SHAPE;
  if (dy_row_stride == 0) dy_row_stride = Cout;
  if (x_row_stride == 0) x_row_stride = Cin;
  if (dy_row_stride < Cout || x_row_stride < Cin || dy_row_stride % 8 || x_row_stride % 8) return HD_ERR_SHAPE;
  if (!hd_aligned16(dy) || !hd_aligned16(x) || !hd_aligned16(dw) || !hd_aligned16(workspace)) return HD_ERR_ALIGN;
  const Wgrad3Plan p = wgrad3_plan(B, H, Cin, Cout, W, stride);
  if (workspace_bytes < (size_t)p.partials * Cout * 9 * Cin * sizeof(float)) return HD_ERR_WORKSPACE;
  hdfuse::ReduceJob jv;
  const hd_status js = make_job_view(piggyback, &jv);
  if (js != HD_OK) return js;
  hipStream_t s = (hipStream_t)stream;
  const auto* a = static_cast<const unsigned short*>(dy);
  const auto* b = static_cast<const unsigned short*>(x);
  float* slab = static_cast<float*>(workspace);
  const dim3 grid((unsigned)p.tiles, (unsigned)p.partials + piggyback_rows(jv, (unsigned)p.tiles));
  const int WP = (W + 15) / 16 * 16;
  const size_t ring = (size_t)(3 * stride * (WP + 2) + WP) * (64 * 2 + 64);
  const size_t lds = ring;
  const dim3 block(256);
#define HD_WG3(WW, KERNEL)                                                                                         \
  (void)hd_launch(KERNEL<WW>, grid, block, lds, s, a, dy_row_stride, b, x_row_stride, B, H, Cout, Cin, p.rows_per_slice, \
                  p.images_per_slice, slab, p.slices, jv)
#define HD_WGRAD3_CASE(WW)                                                                          \
  case WW:                                                                                          \
    if (stride == 2) HD_WG3(WW, conv3x3s2_wgrad_kernel); else HD_WG3(WW, conv3x3_wgrad_kernel);     \
    break;
  switch (W) {
    HD_WGRAD3_CASE(20)
    HD_WGRAD3_CASE(26)
    HD_WGRAD3_CASE(40)
    HD_WGRAD3_CASE(52)
    HD_WGRAD3_CASE(80)
    HD_WGRAD3_CASE(104)
    default: return HD_ERR_SHAPE;
  }
#undef HD_WGRAD3_CASE
#undef HD_WG3
  HD_RETURN_IF_LAUNCH_FAILED();
  job_out->slab = slab;
  job_out->slices = p.partials;
  job_out->count = (int64_t)Cout * 9 * Cin;
  job_out->dw = dw;
  return HD_OK;
}
