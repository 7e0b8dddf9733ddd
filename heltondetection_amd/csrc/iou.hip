// iou.hip — batched pairwise IoU for gfx950 (SURVEY.md §8a "Batched pairwise IoU").
//
// Two kernels, both memory-layout driven (no MFMA: there is no inner product here):
//
//  pairwise_iou_kernel   materialises out[B,N,M].  HBM-write bound: 4*N*M bytes out against
//                        16*(N+M) in.  Each lane owns 4 consecutive outputs of the flat [N*M]
//                        row space and stores them with one 16-byte store, so a wave writes
//                        1 KiB contiguous per store instruction.  The b boxes (ground truth)
//                        and their areas are staged once per workgroup in LDS and read back
//                        as wave-uniform broadcasts; a boxes come through L1 (4 consecutive
//                        outputs almost always share one a box).
//
//  iou_assign_kernel     fused max/argmax over M per a box and max/argmax over N per b box,
//                        never writing [B,N,M] (16(N+M)+8N algorithmic bytes).  One lane per a
//                        box (16-byte coalesced load), b boxes broadcast from LDS.  Pairs that
//                        do not overlap are rejected after 8 VALU ops (two min, two max, two
//                        sub, two compares) — the IEEE division runs only for overlapping
//                        pairs, which keeps the kernel off the VALU roof at large M.  The
//                        per-b reduction packs (iou bits, ~n) into a 64-bit key: LDS
//                        atomicMax per workgroup (guarded by a plain read, so only improving
//                        lanes issue the atomic), then one global atomicMax per b per
//                        workgroup.
//
// Bit-exactness: hd_iou_full() evaluates the oracle's expression order with IEEE division
// (hipcc's default) and no contraction, so values and argmax indices equal the CPU checker's.
#include <algorithm>

#include "hd_common.h"

namespace {

constexpr int kIouThreads = 256;
constexpr int kIouIters = 8;          // 256 lanes * 4 outputs * 8 = 8192 outputs per workgroup
constexpr int kIouLdsMaxM = 48;       // the flat kernel only runs for M < 48 (hd_pairwise_iou): < 1 KiB of LDS,
                                      // not the 60 KiB that limited a CU to two workgroups

template <bool VEC4>
__global__ __launch_bounds__(kIouThreads) void pairwise_iou_kernel(
    const float4* __restrict__ a, int64_t a_bstride4, const float4* __restrict__ b,
    const int32_t* __restrict__ m_valid, int N, int M, float* __restrict__ out) {
  __shared__ float4 s_box[kIouLdsMaxM];
  __shared__ float s_area[kIouLdsMaxM];
  const int img = blockIdx.y;
  const int mv = m_valid ? min(m_valid[img], M) : M;
  const float4* bb = b + (int64_t)img * M;
  int my_safe = 1;
  for (int m = threadIdx.x; m < M; m += kIouThreads) {
    float4 v = bb[m];
    s_box[m] = v;
    s_area[m] = hd_box_area(v);
    my_safe &= (int)hd_box_div_safe_nn(v);
  }
  // every b box of the image inside the exact range of the division-free quotient (hd_common.h)?
  const bool b_safe = __syncthreads_and(my_safe) != 0;
  const int64_t T = (int64_t)N * M;
  const float4* aa = a + (int64_t)img * a_bstride4;
  float* o = out + (int64_t)img * T;
  const int64_t span0 = (int64_t)blockIdx.x * (kIouThreads * 4 * kIouIters);
#pragma unroll 1
  for (int it = 0; it < kIouIters; ++it) {
    const int64_t e0 = span0 + ((int64_t)it * kIouThreads + threadIdx.x) * 4;
    if (e0 >= T) break;
    int n = (int)(e0 / M);
    int m = (int)(e0 - (int64_t)n * M);
    float4 ab = aa[n];
    float aarea = hd_box_area(ab);
    bool fast = b_safe && hd_box_div_safe_pos(ab);
    uint32_t ones = 0u;
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (e0 + k < T) {
        if (m < mv) {
          float4 gb = s_box[m];
          float garea = s_area[m];
          r[k] = fast ? hd_iou_fast(ab, aarea, gb, garea, &ones) : hd_iou_full(ab, aarea, gb, garea);
          if (__builtin_expect(hd_ones_acc(ones), 0)) {  // all-ones denominator: the IEEE division decides
            r[k] = hd_iou_full(ab, aarea, gb, garea);
            ones = 0u;
          }
        } else {
          r[k] = -1.0f;
        }
        if (++m == M) {
          m = 0;
          ++n;
          if (n < N) {
            ab = aa[n];
            aarea = hd_box_area(ab);
            fast = b_safe && hd_box_div_safe_pos(ab);
          }
        }
      } else {
        r[k] = 0.0f;
      }
    }
    if (VEC4) {
      *reinterpret_cast<float4*>(o + e0) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (e0 + k < T) o[e0 + k] = r[k];
    }
  }
}

// Lane-owns-columns variant (M >= 48).  A lane keeps C consecutive b boxes (C = 4, 2 or 1) and
// their areas in registers; a wave therefore spans 64*C columns and emits one C*4-byte store per
// lane per row: with C = 4 that is a 16-byte store, 1 KiB contiguous per wave.  The 32 a boxes
// of the wave's row sub-tile are fetched with one coalesced vector load (lane r holds row r) and
// broadcast with v_readlane inside a branch-free unrolled loop, so the inner loop has no memory
// latency, no LDS access and no index arithmetic.  Measured (scratch micro-benchmark, B=2
// N=409200 M=512): C=1 3.9 TB/s (VALU-bound on the IEEE division), C=4 5.6 TB/s, against 6.3 TB/s
// for the same stores with no arithmetic and 6.5 TB/s for hipMemset.

__device__ __forceinline__ float hd_readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <int C>
struct ColStore;
template <>
struct ColStore<4> {
  static __device__ __forceinline__ void st(float* p, const float* v) {
    // plain stores: non-temporal ones were measured 20 % slower at M = 500, where the lines that straddle two
    // rows are completed in L2 by the next row's store
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <>
struct ColStore<2> {
  static __device__ __forceinline__ void st(float* p, const float* v) {
    *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
  }
};
template <>
struct ColStore<1> {
  static __device__ __forceinline__ void st(float* p, const float* v) { *p = v[0]; }
};

// NCH = number of 64*C-column spans a lane keeps in registers at once.  With NCH spans covering
// the whole row (M <= 64*C*NCH) the loop order is rows-outer: a wave writes each output row
// completely before moving on, so the cache lines that straddle two rows (M*4 not a multiple of
// 128 B) are completed within a few instructions instead of by a later pass — worth 3.2 -> 4.5 TB/s
// at M = 500.  Wider M falls back to span-outer order (NCH = 1, loop over spans).
template <int C, int NCH>
__global__ __launch_bounds__(256) void pairwise_iou_cols_kernel(const float4* __restrict__ a, int64_t a_bstride4,
                                                                const float4* __restrict__ b,
                                                                const int32_t* __restrict__ m_valid, int N, int M,
                                                                int rpw, float* __restrict__ out) {
  const int img = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mv = m_valid ? min(m_valid[img], M) : M;
  const float4* __restrict__ aa = a + (int64_t)img * a_bstride4;
  const float4* __restrict__ bb = b + (int64_t)img * M;
  float* __restrict__ o = out + (int64_t)img * N * M;
  // rpw rows per wave (8, 16 or 32; the host picks fewer rows per wave when the grid would otherwise run as only a
  // handful of rounds of resident waves: the last, partly filled round then costs a whole tile time)
  const int base = (blockIdx.x * 4 + wave) * rpw;
  const int nr = max(0, min(rpw, N - base));
  if (nr == 0) return;
  const float4 rb = lane < nr ? aa[base + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
  const float ra = hd_box_area(rb);
  const bool rows_unsafe = lane < nr && !hd_box_div_safe_pos(rb);
  constexpr int span = 64 * C;
  for (int s0 = 0; s0 < M; s0 += span * NCH) {
    float4 g[NCH][C];
    float ga[NCH][C];
    bool ok[NCH];
    bool unsafe = rows_unsafe;
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const int m0 = s0 + h * span + lane * C;
      ok[h] = m0 < M;  // M % C == 0, so a lane's C columns are all inside or all outside
#pragma unroll
      for (int c = 0; c < C; ++c) g[h][c] = bb[ok[h] ? m0 + c : 0];  // unconditional: all NCH*C loads in flight at once
    }
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        if (!ok[h]) g[h][c] = make_float4(0.f, 0.f, 0.f, 0.f);
        ga[h][c] = hd_box_area(g[h][c]);
        unsafe |= !hd_box_div_safe_nn(g[h][c]);
      }
    }
    // every box of this wave's tile inside the exact range of the division-free quotient (hd_common.h)?
    const bool tile_fast = __ballot(unsafe) == 0ull;
    float* __restrict__ oc = o + (int64_t)base * M + s0 + lane * C;
    // the common case (full 32-row sub-tile, no padded columns in reach) runs without the two
    // per-output predicates: the kernel is VALU-bound, every instruction per output counts
    const bool plain = (nr == rpw) && (s0 + span * NCH <= mv || M == mv);
    if (plain && tile_fast) {
      // the common case on real data: packed-f32 quotient, ~3 issue slots per output instead of ~13 for the IEEE
      // division; same bits (tests/proofs/iou_div_proof.c).  A row whose outputs met an all-ones denominator
      // (detector accumulated over the lane's outputs, one ballot per row) is recomputed with hd_iou_full.
#pragma unroll 8
      for (int r = 0; r < rpw; ++r) {
        const float4 ab = make_float4(hd_readlane_f(rb.x, r), hd_readlane_f(rb.y, r), hd_readlane_f(rb.z, r),
                                      hd_readlane_f(rb.w, r));
        const float aar = hd_readlane_f(ra, r);
        float v[NCH][C];
        uint32_t ones = 0u;
#pragma unroll
        for (int h = 0; h < NCH; ++h) {
          if (C >= 2) {
#pragma unroll
            for (int c = 0; c + 1 < C; c += 2) {
              const hd_f2 q = hd_iou_fast2(ab, aar, g[h][c], ga[h][c], g[h][c + 1], ga[h][c + 1], &ones);
              v[h][c] = q.x;
              v[h][c + 1] = q.y;
            }
          } else {
            v[h][0] = hd_iou_fast(ab, aar, g[h][0], ga[h][0], &ones);
          }
        }
        if (__builtin_expect(__ballot(hd_ones_acc(ones)) != 0ull, 0)) {
#pragma unroll
          for (int h = 0; h < NCH; ++h)
#pragma unroll
            for (int c = 0; c < C; ++c) v[h][c] = hd_iou_full(ab, aar, g[h][c], ga[h][c]);
        }
#pragma unroll
        for (int h = 0; h < NCH; ++h)
          if (ok[h]) ColStore<C>::st(oc + (int64_t)r * M + h * span, v[h]);
      }
    } else if (plain) {
#pragma unroll 8
      for (int r = 0; r < rpw; ++r) {
        const float4 ab = make_float4(hd_readlane_f(rb.x, r), hd_readlane_f(rb.y, r), hd_readlane_f(rb.z, r),
                                      hd_readlane_f(rb.w, r));
        const float aar = hd_readlane_f(ra, r);
#pragma unroll
        for (int h = 0; h < NCH; ++h) {
          float v[C];
#pragma unroll
          for (int c = 0; c < C; ++c) v[c] = hd_iou_full(ab, aar, g[h][c], ga[h][c]);
          if (ok[h]) ColStore<C>::st(oc + (int64_t)r * M + h * span, v);
        }
      }
    } else {
#pragma unroll 4
      for (int r = 0; r < rpw; ++r) {
        const float4 ab = make_float4(hd_readlane_f(rb.x, r), hd_readlane_f(rb.y, r), hd_readlane_f(rb.z, r),
                                      hd_readlane_f(rb.w, r));
        const float aar = hd_readlane_f(ra, r);
#pragma unroll
        for (int h = 0; h < NCH; ++h) {
          float v[C];
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const float x = hd_iou_full(ab, aar, g[h][c], ga[h][c]);
            v[c] = (s0 + h * span + lane * C + c) < mv ? x : -1.0f;
          }
          if (ok[h] && r < nr) ColStore<C>::st(oc + (int64_t)r * M + h * span, v);
        }
      }
    }
  }
}

// Small-M variant (M / C <= 32): several ROWS share one wave instruction.  A lane owns C consecutive columns
// (its C b boxes and their areas live in registers for the whole kernel) of row `lane / (M/C)` of the current
// group of rpi = 64 / (M/C) rows, so a wave store still covers rpi*M*4 contiguous bytes (960 B at M = 40) and
// there is no per-output index arithmetic, no LDS and no box that changes in the middle of a lane's vector —
// the flat kernel spends ~30 instruction slots per output on exactly those, this one ~15.
// A wave takes 64 rows at a time: their boxes are loaded once, coalesced, for the range check of the
// division-free quotient; the per-group row loads that follow hit L1.
constexpr int kRowsTile = 64;

template <int C>
__global__ __launch_bounds__(256) void pairwise_iou_rows_kernel(const float4* __restrict__ a, int64_t a_bstride4,
                                                                const float4* __restrict__ b,
                                                                const int32_t* __restrict__ m_valid, int N, int M,
                                                                float* __restrict__ out) {
  const int img = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int base = (blockIdx.x * 4 + wave) * kRowsTile;
  if (base >= N) return;  // whole wave
  const int mv = m_valid ? min(m_valid[img], M) : M;
  const float4* __restrict__ aa = a + (int64_t)img * a_bstride4;
  const float4* __restrict__ bb = b + (int64_t)img * M;
  float* __restrict__ o = out + (int64_t)img * N * M;
  const int cgn = M / C;     // column groups per row (<= 32)
  const int rpi = 64 / cgn;  // rows per wave instruction (>= 2)
  const int lrow = lane / cgn, lcg = lane - lrow * cgn;
  const bool active = lrow < rpi;
  float4 g[C];
  float ga[C];
  bool unsafe = false;
#pragma unroll
  for (int c = 0; c < C; ++c) g[c] = bb[lcg * C + c];  // lcg < cgn for every lane: always in range
#pragma unroll
  for (int c = 0; c < C; ++c) {
    ga[c] = hd_box_area(g[c]);
    unsafe |= !hd_box_div_safe_nn(g[c]);
  }
  const int nr = min(kRowsTile, N - base);
  const float4 tb = aa[base + min(lane, nr - 1)];
  unsafe |= !hd_box_div_safe_pos(tb);
  // fast: no padded column, every b box and every row box of this tile inside the exact range of the quotient
  const bool fast = (mv == M) && __ballot(unsafe) == 0ull;
  for (int r0 = 0; r0 < nr; r0 += rpi) {
    const int rr = r0 + lrow;
    const bool on = active && rr < nr;
    const int n = base + (on ? rr : 0);
    const float4 ab = aa[n];
    const float aar = hd_box_area(ab);
    float v[C];
    if (fast) {
      uint32_t ones = 0u;
      if (C >= 2) {
#pragma unroll
        for (int c = 0; c + 1 < C; c += 2) {
          const hd_f2 q = hd_iou_fast2(ab, aar, g[c], ga[c], g[c + 1], ga[c + 1], &ones);
          v[c] = q.x;
          v[c + 1] = q.y;
        }
      } else {
        v[0] = hd_iou_fast(ab, aar, g[0], ga[0], &ones);
      }
      if (__builtin_expect(__ballot(hd_ones_acc(ones)) != 0ull, 0)) {  // all-ones denominator: IEEE division
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = hd_iou_full(ab, aar, g[c], ga[c]);
      }
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) v[c] = (lcg * C + c) < mv ? hd_iou_full(ab, aar, g[c], ga[c]) : -1.0f;
    }
    if (on) ColStore<C>::st(o + (int64_t)n * M + lcg * C, v);
  }
}

// ---------------------------------------------------------------------------------------------

constexpr int kAssignThreads = 256;
constexpr int kAssignLdsMaxM = 2048;  // 2048 * (16 + 4 + 8) B = 56 KiB

__device__ __forceinline__ unsigned long long pack_key(float iou, int n) {
  return ((unsigned long long)__float_as_uint(iou) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)n);
}

// one pair for the assignment kernel (runs only for overlapping pairs): fast quotient when both boxes qualify,
// IEEE division otherwise and for an all-ones denominator
__device__ __forceinline__ float iou_pair(bool fast, const float4 a, float area_a, const float4 b, float area_b) {
  if (fast) {
    uint32_t ones = 0u;
    const float v = hd_iou_fast(a, area_a, b, area_b, &ones);
    if (__builtin_expect(!hd_ones_acc(ones), 1)) return v;
  }
  return hd_iou_full(a, area_a, b, area_b);
}

template <bool WITH_GT>
__global__ __launch_bounds__(kAssignThreads) void iou_assign_kernel(
    const float4* __restrict__ a, int64_t a_bstride4, const float4* __restrict__ b,
    const int32_t* __restrict__ m_valid, int N, int M, int m_chunk0, int lds_cap, float* __restrict__ max_iou,
    int32_t* __restrict__ argmax, unsigned long long* __restrict__ gt_key) {
  // One pass handles b boxes [m_chunk0, m_chunk0 + kAssignLdsMaxM); the host loops chunks for
  // M > kAssignLdsMaxM and the kernel then continues the running max held in max_iou/argmax.
  // Dynamic LDS sized by the host to the b boxes one pass really stages (lds_cap = min(M, kAssignLdsMaxM)
  // rounded up to 4): with the static 56 KiB of the largest case only two workgroups fit a CU, which halved
  // the throughput of this LDS-latency-bound loop for the common M = 20..500.
  extern __shared__ __align__(16) unsigned char s_raw[];
  float4* s_box = reinterpret_cast<float4*>(s_raw);
  float* s_area = reinterpret_cast<float*>(s_box + lds_cap);
  unsigned long long* s_key = reinterpret_cast<unsigned long long*>(s_area + lds_cap);  // WITH_GT only
  const int img = blockIdx.y;
  const int mv = m_valid ? min(m_valid[img], M) : M;
  const int m_end = min(mv, m_chunk0 + kAssignLdsMaxM);
  const int mc = max(m_end - m_chunk0, 0);
  const float4* bb = b + (int64_t)img * M + m_chunk0;
  int my_safe = 1;
  for (int m = threadIdx.x; m < mc; m += kAssignThreads) {
    float4 v = bb[m];
    s_box[m] = v;
    s_area[m] = hd_box_area(v);
    my_safe &= (int)hd_box_div_safe_nn(v);
    if (WITH_GT) s_key[m] = 0ull;
  }
  const bool b_safe = __syncthreads_and(my_safe) != 0;  // division-free exact quotient allowed (hd_common.h)
  const float4* aa = a + (int64_t)img * a_bstride4;
  for (int n = blockIdx.x * kAssignThreads + threadIdx.x; n < N; n += gridDim.x * kAssignThreads) {
    const float4 ab = aa[n];
    const float aarea = hd_box_area(ab);
    const bool fast = b_safe && hd_box_div_safe_pos(ab);
    float best;
    int bidx;
    if (m_chunk0 == 0) {
      // oracle: the first valid b always wins over the initial -1 (IoU >= 0)
      best = mv > 0 ? 0.0f : -1.0f;
      bidx = mv > 0 ? 0 : -1;
      if (mv > 0) {
        // exact value for m = 0 (may be > 0)
        best = iou_pair(fast, ab, aarea, s_box[0], s_area[0]);
        if (WITH_GT && best > 0.0f) {
          unsigned long long key = pack_key(best, n);
          if (key > s_key[0]) atomicMax(&s_key[0], key);
        }
      }
    } else {
      best = max_iou[(int64_t)img * N + n];
      bidx = argmax[(int64_t)img * N + n];
    }
    // 4 b boxes per iteration: the four LDS reads and overlap tests are independent, so the loop is no
    // longer a chain of (LDS latency -> test -> branch) per pair; the exact IoU (with its IEEE
    // division) still runs only for overlapping pairs, in index order so ties keep the lowest m.
    int m = (m_chunk0 == 0 ? 1 : 0);
    for (; m + 3 < mc; m += 4) {
      const float4 g0 = s_box[m], g1 = s_box[m + 1], g2 = s_box[m + 2], g3 = s_box[m + 3];
      const bool o0 = hd_overlaps(ab, g0), o1 = hd_overlaps(ab, g1), o2 = hd_overlaps(ab, g2),
                 o3 = hd_overlaps(ab, g3);
      if (!(o0 || o1 || o2 || o3)) continue;
#define HD_PAIR(OK, G, MM)                                                   \
  if (OK) {                                                                  \
    const float v = iou_pair(fast, ab, aarea, G, s_area[MM]);                \
    if (v > best) {                                                          \
      best = v;                                                              \
      bidx = m_chunk0 + (MM);                                                \
    }                                                                        \
    if (WITH_GT && v > 0.0f) {                                               \
      const unsigned long long key = pack_key(v, n);                         \
      if (key > s_key[MM]) atomicMax(&s_key[MM], key);                       \
    }                                                                        \
  }
      HD_PAIR(o0, g0, m)
      HD_PAIR(o1, g1, m + 1)
      HD_PAIR(o2, g2, m + 2)
      HD_PAIR(o3, g3, m + 3)
    }
    for (; m < mc; ++m) {
      const float4 gb = s_box[m];
      const bool ok = hd_overlaps(ab, gb);
      HD_PAIR(ok, gb, m)
    }
#undef HD_PAIR
    max_iou[(int64_t)img * N + n] = best;
    argmax[(int64_t)img * N + n] = bidx;
  }
  if (WITH_GT) {
    __syncthreads();
    unsigned long long* gk = gt_key + (int64_t)img * M + m_chunk0;
    for (int m = threadIdx.x; m < mc; m += kAssignThreads) {
      // Keys only grow, so a relaxed (possibly stale) read can only let a redundant atomic through, never
      // hide an improving one.  Without the guard every workgroup sends M atomics onto the same few cache
      // lines of its image (2048 workgroups x 500 b boxes: ~0.3 ms of serialised atomics at config 5).
      unsigned long long k = s_key[m];
      if (k != 0ull && k > __hip_atomic_load(&gk[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&gk[m], k);
    }
  }
}

__global__ void iou_assign_finalize_kernel(const unsigned long long* __restrict__ gt_key,
                                           const int32_t* __restrict__ m_valid, int B, int N, int M,
                                           float* __restrict__ gt_max_iou,
                                           int32_t* __restrict__ gt_argmax) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * M) return;
  int img = i / M, m = i - img * M;
  int mv = m_valid ? min(m_valid[img], M) : M;
  float v = -1.0f;
  int idx = -1;
  if (m < mv && N > 0) {
    unsigned long long k = gt_key[i];
    if (k == 0ull) {  // every IoU with this box was 0: the oracle keeps the first a box
      v = 0.0f;
      idx = 0;
    } else {
      v = __uint_as_float((uint32_t)(k >> 32));
      idx = (int)(0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull));
    }
  }
  gt_max_iou[i] = v;
  gt_argmax[i] = idx;
}

}  // namespace

extern "C" hd_status hd_pairwise_iou(const float* a, int64_t a_bstride, const float* b,
                                     const int32_t* m_valid, int B, int N, int M, float* out,
                                     hd_stream stream) {
  if (B < 0 || N < 0 || M < 0 || (a_bstride & 3)) return HD_ERR_SHAPE;
  if (B == 0 || N == 0 || M == 0) return HD_OK;
  if (!a || !b || !out) return HD_ERR_NULL;
  if (!hd_aligned16(a) || !hd_aligned16(b)) return HD_ERR_ALIGN;
  const int64_t T = (int64_t)N * M;
  if (B > 65535) return HD_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const float4* a4c = reinterpret_cast<const float4*>(a);
  const float4* b4c = reinterpret_cast<const float4*>(b);
  // small M: several rows per wave instruction (M / C <= 32 column groups, C = 4 or 2 columns per lane)
  {
    int C = 0;
    if (M % 4 == 0 && M / 4 <= 32 && hd_aligned16(out)) C = 4;
    else if (M % 2 == 0 && M / 2 <= 32 && (reinterpret_cast<uintptr_t>(out) & 7u) == 0) C = 2;
    if (C != 0) {
      const int64_t gx = ((int64_t)N + 4 * kRowsTile - 1) / (4 * kRowsTile);
      dim3 rgrid((unsigned)gx, (unsigned)B);
      if (C == 4)
        (void)hd_launch(pairwise_iou_rows_kernel<4>, rgrid, dim3(256), 0, s, a4c, a_bstride / 4, b4c, m_valid, N, M,
                           out);
      else
        (void)hd_launch(pairwise_iou_rows_kernel<2>, rgrid, dim3(256), 0, s, a4c, a_bstride / 4, b4c, m_valid, N, M,
                           out);
      HD_RETURN_IF_LAUNCH_FAILED();
      return HD_OK;
    }
  }
  if (M >= 48) {
    // columns per lane: the widest store whose last 64*C-column span is still >= 70 % occupied
    auto util = [&](int C) { const int span = 64 * C; return (double)M / (double)(((M + span - 1) / span) * span); };
    int C = 1;
    if (M % 4 == 0 && hd_aligned16(out) && util(4) >= 0.7) C = 4;
    else if (M % 2 == 0 && (reinterpret_cast<uintptr_t>(out) & 7u) == 0 && util(2) >= 0.7) C = 2;
    const int spans = (M + 64 * C - 1) / (64 * C);
    const int nch = spans <= 3 ? spans : 1;  // rows-outer when the whole row fits in registers
    // (tried and dropped: one 64*C-column span per wave with 2 / 4 waves sharing a row tile — half the registers, twice
    // the waves in flight — measured the same 5.0-5.5 TB/s: the kernel is on the store path's side of the roofline)
    // rows per wave: 32 when that still gives >= ~8 rounds of resident waves (16 per CU x 256 CUs), else 16 or 8
    // (measured: 16 x 25200 x 512 runs 5.0 / 5.3 / 5.6 TB/s at 32 / 16 / 8 rows, 2 x 409200 x 500 prefers 16-32)
    int rpw = 32;
    while (rpw > 8 && (int64_t)B * ((N + rpw - 1) / rpw) < 8ll * 4096) rpw >>= 1;
    dim3 cgrid((unsigned)((N + 4 * rpw - 1) / (4 * rpw)), (unsigned)B);
#define HD_LAUNCH_COLS(CC, NN)                                                                                      \
  (void)hd_launch(pairwise_iou_cols_kernel<CC, NN>, cgrid, dim3(256), 0, s, a4c, a_bstride / 4, b4c, m_valid, N, \
                     M, rpw, out)
    if (C == 4) {
      if (nch == 1) HD_LAUNCH_COLS(4, 1);
      else if (nch == 2) HD_LAUNCH_COLS(4, 2);
      else HD_LAUNCH_COLS(4, 3);
    } else if (C == 2) {
      if (nch == 1) HD_LAUNCH_COLS(2, 1);
      else if (nch == 2) HD_LAUNCH_COLS(2, 2);
      else HD_LAUNCH_COLS(2, 3);
    } else {
      if (nch == 1) HD_LAUNCH_COLS(1, 1);
      else if (nch == 2) HD_LAUNCH_COLS(1, 2);
      else HD_LAUNCH_COLS(1, 3);
    }
#undef HD_LAUNCH_COLS
    HD_RETURN_IF_LAUNCH_FAILED();
    return HD_OK;
  }
  const int64_t per_block = (int64_t)kIouThreads * 4 * kIouIters;
  const int64_t gx = (T + per_block - 1) / per_block;
  if (gx > 0x7fffffff) return HD_ERR_SHAPE;
  dim3 grid((unsigned)gx, (unsigned)B);
  const bool vec = (T % 4 == 0) && hd_aligned16(out);
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  static_assert(kIouLdsMaxM >= 48, "the flat kernel stages all M < 48 b boxes in LDS");
#define HD_LAUNCH_PIOU(V) \
  hipLaunchKernelGGL((pairwise_iou_kernel<V>), grid, dim3(kIouThreads), 0, s, a4, a_bstride / 4, b4, m_valid, N, M, out)
  if (vec) HD_LAUNCH_PIOU(true);
  else HD_LAUNCH_PIOU(false);
#undef HD_LAUNCH_PIOU
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" size_t hd_iou_assign_workspace_bytes(int B, int N, int M) {
  (void)N;
  if (B <= 0 || M <= 0) return 0;
  return (size_t)B * (size_t)M * sizeof(unsigned long long);
}

extern "C" hd_status hd_iou_assign(const float* a, int64_t a_bstride, const float* b,
                                   const int32_t* m_valid, int B, int N, int M, float* max_iou,
                                   int32_t* argmax, float* gt_max_iou, int32_t* gt_argmax,
                                   void* workspace, size_t workspace_bytes, hd_stream stream) {
  if (B < 0 || N < 0 || M < 0 || (a_bstride & 3)) return HD_ERR_SHAPE;
  if (B == 0) return HD_OK;
  if (B > 65535) return HD_ERR_SHAPE;
  const bool with_gt = gt_max_iou != nullptr || gt_argmax != nullptr;
  if (with_gt && (!gt_max_iou || !gt_argmax)) return HD_ERR_NULL;
  if (N > 0 && (!a || !max_iou || !argmax)) return HD_ERR_NULL;
  if (M > 0 && !b) return HD_ERR_NULL;
  if ((a && !hd_aligned16(a)) || (b && !hd_aligned16(b))) return HD_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  unsigned long long* keys = nullptr;
  if (with_gt && M > 0) {
    if (!workspace || workspace_bytes < hd_iou_assign_workspace_bytes(B, N, M)) return HD_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(workspace) & 7u) return HD_ERR_ALIGN;
    keys = reinterpret_cast<unsigned long long*>(workspace);
    if (hd_fill32(keys, 0u, (size_t)B * M * 2, s) != HD_OK) return HD_ERR_LAUNCH;
  }
  if (N > 0) {
    int gx = (N + kAssignThreads - 1) / kAssignThreads;
    // enough workgroups to fill 256 CUs several times over, grid-stride beyond that
    const int cap = max(1, 4096 / B);
    if (gx > cap) gx = cap;
    dim3 grid((unsigned)gx, (unsigned)B);
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    int chunk = 0;
    const int lds_cap = std::max(4, (std::min(M, kAssignLdsMaxM) + 3) & ~3);
    const size_t lds = (size_t)lds_cap * (sizeof(float4) + sizeof(float) + (with_gt ? sizeof(unsigned long long) : 0));
    do {
      if (with_gt)
        hipLaunchKernelGGL((iou_assign_kernel<true>), grid, dim3(kAssignThreads), lds, s, a4, a_bstride / 4, b4,
                           m_valid, N, M, chunk, lds_cap, max_iou, argmax, keys);
      else
        hipLaunchKernelGGL((iou_assign_kernel<false>), grid, dim3(kAssignThreads), lds, s, a4, a_bstride / 4, b4,
                           m_valid, N, M, chunk, lds_cap, max_iou, argmax, keys);
      HD_RETURN_IF_LAUNCH_FAILED();
      chunk += kAssignLdsMaxM;
    } while (chunk < M);
  }
  if (with_gt && M > 0) {
    int total = B * M;
    hipLaunchKernelGGL(iou_assign_finalize_kernel, dim3((total + 255) / 256), dim3(256), 0, s, keys, m_valid, B, N,
                       M, gt_max_iou, gt_argmax);
    HD_RETURN_IF_LAUNCH_FAILED();
  }
  return HD_OK;
}
