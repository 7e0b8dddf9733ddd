// pointwise_longk.hip — the 1x1 convolution forward for LONG reductions on few rows (Cin = 512 on the 20x20 / 40x40
// levels: the C3 / SPPF / neck layers behind a concatenation), with the following BatchNorm's statistics in the epilogue.
// Reached through hd_pointwise_conv (pointwise_conv.hip), which keeps the streaming form for Cin <= 256.
//
// Why its own kernel: these launches are 3 - 7 GFLOP on 6 400 - 25 600 rows — a 113 x 113 output patch per CU.  Nothing can be
// amortised, so the kernel is built to have no serial phase twice:
//   * a workgroup owns 128 rows x BN columns (BN = 128, or 64 where 128 would leave most CUs without a workgroup) and holds its
//     WHOLE weight slab (BN x K bf16 = 128 / 64 KB) in LDS: staged once, one barrier, none afterwards;
//   * each wave owns 32 rows and pulls them straight from global memory into MFMA A-operand registers (lane l: row l & 31,
//     16-byte chunk 2 s + (l >> 5) of k-step s), ALL K / 16 loads of the row block issued right behind the slab's loads and
//     ahead of the slab's LDS stores, so both operands travel at once;
//   * the B fragments of k-step s + 2 are requested from LDS before the MFMAs of k-step s issue (one wave per SIMD: nobody
//     else hides an LDS round trip);
//   * the result leaves through a wave-private LDS patch as 16-byte stores along full rows; the statistics (of the ROUNDED
//     bf16 values, as everywhere) are folded through LDS into one replica of the slot.
// The round-2 tiled kernel (pointwise_conv_kernel: both operands through LDS in BK = 64 steps, one barrier per step) measured
// 22.8 - 59 us on these shapes against the library's 9.3 - 17.8 us (+ 3.6 - 4.6 us for the separate statistics launch):
// tools/pw_long_k_probe.py, profiles/r05_pw_long_k.md.
#include "hd_common.h"

#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int kStatReplicas = 16;  // == hd_bn_act_replicas(), checked by the caller

template <int K>
struct SlabK {  // [rows][K] bf16; 16-byte chunks XOR-swizzled by the row's low four bits: 16 rows at one chunk index -> 16 bank groups
  static constexpr int CPR = K / 8;
  __device__ static __forceinline__ int off(int row, int chunk) { return row * (K * 2) + ((chunk ^ (row & 15)) << 4); }
};

template <int K, int TN, bool STATS>
__global__ __launch_bounds__(256) void pointwise_longk_kernel(const unsigned short* __restrict__ A, long long lda,
                                                              const unsigned short* __restrict__ Bw,
                                                              unsigned short* __restrict__ Cmat, long long ldc, int M, int N,
                                                              float* __restrict__ stats) {
  constexpr int BN = TN * 32, KS = K / 16, CPR = K / 8;
  constexpr int SLAB_BYTES = BN * K * 2;
  constexpr int PATCH_ROW = 64 * 2 + 16, PATCH_BYTES = 32 * PATCH_ROW;  // 64 columns at a time
  constexpr int HALVES = BN / 64;
  static_assert(BN % 64 == 0, "the epilogue moves 64 columns at a time");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [slab | 4 wave patches | statistics fold]
  float* s_st = reinterpret_cast<float*>(smem + SLAB_BYTES + 4 * PATCH_BYTES);  // [2][BN]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 31, fh = lane >> 5;
  const int NT = N / BN, G = gridDim.x;
  const int lb = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;  // consecutive ids on one XCD
  const int nt = lb % NT, n0 = nt * BN;  // G is a multiple of NT: one column tile per workgroup
  const int RB = (M + 127) / 128;        // row blocks of 128 rows (32 per wave)
  const int rb0 = lb / NT, rstep = G / NT;

  // k-steps whose operand loads are issued AHEAD of the slab's LDS stores; the rest follows the stores, in the registers the slab
  // chunks vacate (128-wide tiles: 32 + 32 sixteen-byte loads in flight would not fit 256 VGPRs, measured as 12 spilled)
  constexpr int KS0 = TN >= 4 ? KS / 2 : KS;
  u32x4 a[KS];
  auto load_rows = [&](int rb, auto first, auto last) {
    int m = rb * 128 + wave * 32 + frow;
    m = m < M ? m : M - 1;  // rows past the end re-read the last row; their results are never stored
    const unsigned short* p = A + (long long)m * lda + fh * 8;
#pragma unroll
    for (int s = decltype(first)::value; s < decltype(last)::value; ++s) a[s] = *reinterpret_cast<const u32x4*>(p + s * 16);
  };
  const std::integral_constant<int, 0> k_begin;
  const std::integral_constant<int, KS0> k_split;
  const std::integral_constant<int, KS> k_end;
  {
    // The slab's BN x K / 8 chunks, ALL of a thread's loads issued first (the weights sit in L2: one round trip), THEN the row
    // block's K / 16 operand loads (HBM): vmcnt retires in issue order, so the LDS stores below wait for the slab only and the
    // operand rows keep travelling through the staging, the barrier and the first k-steps.
    constexpr int NCH = BN * CPR, WIT = NCH / 256;
    static_assert(NCH % 256 == 0, "slab chunks per thread");
    u32x4 wreg[WIT];
#pragma unroll
    for (int p = 0; p < WIT; ++p) {
      const int id = p * 256 + tid;
      wreg[p] = *reinterpret_cast<const u32x4*>(Bw + (long long)(n0 + id / CPR) * K + (id % CPR) * 8);
    }
    __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise interleaves loads and stores ten at a time)
    load_rows(rb0, k_begin, k_split);  // UNCONDITIONAL (rows are clamped): behind a branch the compiler cannot count these loads and makes the
                     // stores below wait for them too (s_waitcnt vmcnt counts down to 0 instead of stopping at K / 16)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < WIT; ++p) {
      const int id = p * 256 + tid;
      *reinterpret_cast<u32x4*>(smem + SlabK<K>::off(id / CPR, id % CPR)) = wreg[p];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (KS0 < KS) load_rows(rb0, k_split, k_end);
    if (STATS)
      for (int i = tid; i < 2 * BN; i += 256) s_st[i] = 0.0f;
  }
  __syncthreads();

  unsigned char* patch = smem + SLAB_BYTES + wave * PATCH_BYTES;
  float s_sum[TN], s_sq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) s_sum[j] = s_sq[j] = 0.0f;

  for (int rb = rb0; rb < RB; rb += rstep) {
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    constexpr int PD = 2;
    bf16x8 fb[PD + 1][TN];
    auto fetch = [&](int s, bf16x8 (&b)[TN]) {
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8*>(smem + SlabK<K>::off(j * 32 + frow, 2 * s + fh));
    };
#pragma unroll
    for (int s = 0; s < PD; ++s) fetch(s, fb[s]);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + PD < KS) fetch(s + PD, fb[(s + PD) % (PD + 1)]);
      __builtin_amdgcn_sched_barrier(0);  // keep the requests above the multiplies they are meant to overlap
      const bf16x8 av = __builtin_bit_cast(bf16x8, a[s]);
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, fb[s % (PD + 1)][j], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int m0 = rb * 128 + wave * 32;
    if (rb + rstep < RB) load_rows(rb + rstep, k_begin, k_end);  // the next block's operand rows, behind this block's multiplies
    // C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)  ->  patch[row][col] as bf16, 64 columns at a time
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * h + jj;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
          const unsigned short v = __builtin_bit_cast(unsigned short, (__bf16)acc[j][r]);  // v_cvt_pk_bf16_f32, RNE
          *reinterpret_cast<unsigned short*>(patch + row * PATCH_ROW + (jj * 32 + frow) * 2) = v;
          if (STATS && m0 + row < M) {
            const float f = __uint_as_float((unsigned)v << 16);
            s_sum[j] += f;
            s_sq[j] = fmaf(f, f, s_sq[j]);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own patch writes, before its lanes read each other's
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // 32 rows x 8 chunks of 16 bytes over 64 lanes
        const int id = q * 64 + lane, row = id >> 3, c = id & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(patch + row * PATCH_ROW + c * 16);
        if (m0 + row < M) *reinterpret_cast<uint4*>(Cmat + (long long)(m0 + row) * ldc + n0 + h * 64 + c * 8) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // patch reads done before it is overwritten
    }
  }

  if (STATS) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float s = s_sum[j] + __shfl_xor(s_sum[j], 32);
      const float q = s_sq[j] + __shfl_xor(s_sq[j], 32);
      if (fh == 0) {
        atomicAdd(&s_st[j * 32 + frow], s);  // four waves per address
        atomicAdd(&s_st[BN + j * 32 + frow], q);
      }
    }
    __syncthreads();
    float* dst = stats + (size_t)(blockIdx.x % kStatReplicas) * 2 * N;
    for (int i = tid; i < 2 * BN; i += 256) atomicAdd(&dst[(i / BN) * N + n0 + (i % BN)], s_st[i]);
  }
}

template <int K, int TN>
hd_status launch_longk(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, int M, int N, float* stats, hipStream_t s) {
  constexpr int BN = TN * 32;
  const int NT = N / BN;
  const long long RB = ((long long)M + 127) / 128;
  long long L = 8;  // the grid is a multiple of 8 (XCD remap) and of NT (one column tile per workgroup)
  while (L % NT) L += 8;
  long long g = RB * NT;
  g = (g + L - 1) / L * L;
  const size_t lds = (size_t)BN * K * 2 + 4 * 32 * (64 * 2 + 16) + 2 * BN * sizeof(float);
  const long long cap = lds > 80 * 1024 ? 256 : 512;  // what is resident at once: one workgroup per CU with a 128 KB slab, two with 64 KB
  if (g > cap) g = cap / L * L;
  if (g < L) g = L;
  const auto* a = static_cast<const unsigned short*>(x);
  const auto* b = static_cast<const unsigned short*>(w);
  auto* c = static_cast<unsigned short*>(y);
  if (stats)
    (void)hd_launch(pointwise_longk_kernel<K, TN, true>, dim3((unsigned)g), dim3(256), lds, s, a, (long long)ldx, b, c, (long long)ldy,
                    M, N, stats);
  else
    (void)hd_launch(pointwise_longk_kernel<K, TN, false>, dim3((unsigned)g), dim3(256), lds, s, a, (long long)ldx, b, c,
                    (long long)ldy, M, N, stats);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

}  // namespace

// Shapes this kernel is preferred on, measured against the library's convolution + the separate statistics launch in replayed
// graphs (tools/pw_long_k_probe.py, profiles/r05_pw_long_k.md): the reduction 512 long and at most 16 384 rows — 6 400 rows:
// 9.2 against 12.2 us (-> 256 channels) and 12.6 against 16.8 us (-> 512); 10 816 rows: 20.4 against 20.3; from 25 600 rows on
// (two row blocks per workgroup) it ties or loses (21.5 against 21.6 us) and the library keeps those.
int hd_pointwise_longk_supported(int64_t R, int Cin, int Cout) {
  if (R < 1 || R > 16384) return 0;
  if (Cin == 512) return Cout % 128 == 0 && Cout / 128 <= 32;
  return 0;  // Cin = 1024 (SPPF cv2) measured slower than the library with its separate statistics launch (25.4 against 21.8 us)
}

hd_status hd_pointwise_longk(const void* x, int64_t ldx, const void* w, int64_t R, int Cin, int Cout, void* y, int64_t ldy,
                             float* stats, hipStream_t s) {
  if (!hd_pointwise_longk_supported(R, Cin, Cout)) return HD_ERR_SHAPE;
  {
    // 128-wide column tiles unless they leave most CUs without a workgroup (512 -> 256 on 6 400 rows: 100 workgroups)
    const long long wide = ((R + 127) / 128) * (Cout / 128);
    return wide >= 160 ? launch_longk<512, 4>(x, ldx, w, y, ldy, (int)R, Cout, stats, s)
                       : launch_longk<512, 2>(x, ldx, w, y, ldy, (int)R, Cout, stats, s);
  }
}
