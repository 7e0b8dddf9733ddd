// hd_nms_core.h — the exact greedy-NMS engine shared by hd_nms_batched, hd_yolo_nms and
// hd_rpn_proposals, plus the workgroup bitonic sort they (and hd_wbf) use.
//
// Pipeline per image (all sizes data-dependent, read from device memory — no host sync):
//   0. an op-specific "prepare" kernel appends one 64-bit key per candidate:
//        key = (order-preserving bits of the score << 32) | (0xFFFFFFFF - candidate index)
//      so a descending sort gives "score descending, ties -> lower index" (DESIGN.md §2.3);
//      candidate boxes / classes stay in index-addressed arrays.
//   1. nms_sort_gather_kernel   one 1024-lane workgroup per image: zero-pad to a power of two,
//      bitonic sort (LDS tiles of 4096 keys, global passes only for strides >= 4096), clamp to
//      max_candidates, gather boxes/classes into sorted order (coalesced for step 2).
//   2. nms_mask_kernel          64x64 tiles of the upper triangle, walked persistently (the tile-pair
//      count comes from the device-side candidate count).  A wave holds 64 column boxes (one per
//      lane), the 64 row boxes are staged in LDS and broadcast; the suppression test of one row
//      against the 64 columns is a single v_cmp whose 64-bit result IS the mask word (the wavefront
//      ballot).  Non-overlapping or other-class pairs exit after ~8 VALU ops.
//   3. nms_scan_kernel          one workgroup per image; lane t owns word t of the removed-set.
//      Per 64-box chunk wave 0 resolves the chunk against its diagonal word with scalar
//      bit-ops, then all lanes OR the kept rows' words (independent loads).  Stops at max_out.
//
// Steps 2 and 3 run in WINDOWS over the score-sorted candidates (first 1024, then up to 4096, 16384, 65536): the
// scan stops as soon as max_out boxes are kept, and everything a later window would have computed — the bulk of the
// N^2/2 pair tests when max_out is a few hundred — is never launched into (the kernels of the remaining windows find
// the image's `done` flag set and exit).  A window's mask kernel computes the tiles whose COLUMN tile lies in the
// window (all rows above the diagonal); its scan first folds the rows kept in earlier windows into the new columns'
// removed-words, then continues the greedy resolve.  State between windows (kept count, kept bits per chunk, done
// flag) lives in the workspace.  Greedy NMS evaluated lazily is still greedy NMS:
// the result is bit-identical to the serial algorithm in oracle/hd_oracle.c: nms_one().
#pragma once
#include <atomic>

#include "hd_common.h"

namespace hdnms {

constexpr int kSortThreads = 1024;
constexpr int kSortTile = 16384;  // keys per LDS tile: 128 KiB of dynamic LDS (the CU has 160 KiB)
constexpr int kMaxCand = 65536;
constexpr int kMaxDevices = 64;  // device ordinals with a cached "LDS opt-in done" flag (others re-issue it)   // 1024 lanes x one 64-bit removed-word each

typedef unsigned long long u64;

__device__ __forceinline__ u64 make_key(float score, uint32_t index) {
  return ((u64)hd_f2key(score) << 32) | (u64)(0xFFFFFFFFu - index);
}
__device__ __forceinline__ uint32_t key_index(u64 k) { return 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull); }
__device__ __forceinline__ float key_score(u64 k) { return hd_key2f((uint32_t)(k >> 32)); }

// compare-exchange steps j = j_from, j_from/2, ..., 1 of bitonic stage k on an LDS tile whose
// first element has global index `base`; final order is DESCENDING.
// A step whose exchange distance j fits inside a wave's own contiguous chunk of the tile (2j <= chunk) only
// moves keys within that chunk, so the wave runs it alone and a wave-level barrier suffices; a workgroup
// barrier is needed only around the steps that cross chunks.  For a 16384-key tile on 16 waves that is 10 of
// the 105 steps.
__device__ __forceinline__ void bitonic_local(u64* lds, int tile_n, int base, int k, int j_from) {
  const int n_waves = (int)(blockDim.x >> 6);
  const int wave = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
  const int chunk = tile_n / n_waves;          // keys per wave
  const bool can_split = chunk >= 128;         // >= 64 pairs per wave-local step: every lane has work
  bool dirty = false;                          // wave-local writes not yet published to the workgroup
  for (int j = j_from; j > 0; j >>= 1) {
    if (can_split && 2 * j <= chunk) {
      const int p0 = wave * (chunk >> 1), p1 = p0 + (chunk >> 1);
      for (int p = p0 + lane; p < p1; p += 64) {
        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
        const int q = i | j;
        const bool desc = ((base + i) & k) == 0;
        const u64 a = lds[i], b = lds[q];
        if ((a < b) == desc) {
          lds[i] = b;
          lds[q] = a;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      dirty = true;
    } else {
      if (dirty) __syncthreads();
      for (int p = threadIdx.x; p < tile_n / 2; p += blockDim.x) {
        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
        const int q = i | j;
        const bool desc = ((base + i) & k) == 0;
        const u64 a = lds[i], b = lds[q];
        if ((a < b) == desc) {
          lds[i] = b;
          lds[q] = a;
        }
      }
      __syncthreads();
      dirty = false;
    }
  }
  if (dirty) __syncthreads();
}

// Sorts keys[0..n_pow2) (global memory) descending with the calling workgroup.
// lds must hold min(n_pow2, lds_keys) keys.  Every thread of the workgroup must call this.
__device__ __forceinline__ void block_sort_desc(u64* keys, int n_pow2, u64* lds, int lds_keys = kSortTile) {
  const int tile_n = n_pow2 < lds_keys ? n_pow2 : lds_keys;
  if (n_pow2 <= 1) return;
  // stage 1: every tile fully sorted for k <= tile_n (alternating direction by tile index)
  for (int base = 0; base < n_pow2; base += tile_n) {
    for (int i = threadIdx.x; i < tile_n; i += blockDim.x) lds[i] = keys[base + i];
    __syncthreads();
    for (int k = 2; k <= tile_n; k <<= 1) bitonic_local(lds, tile_n, base, k, k >> 1);
    for (int i = threadIdx.x; i < tile_n; i += blockDim.x) keys[base + i] = lds[i];
    __syncthreads();
  }
  // stage 2: merges wider than a tile
  for (int k = tile_n << 1; k <= n_pow2; k <<= 1) {
    for (int j = k >> 1; j >= tile_n; j >>= 1) {
      for (int p = threadIdx.x; p < n_pow2 / 2; p += blockDim.x) {
        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
        const int q = i | j;
        const bool desc = (i & k) == 0;
        const u64 a = keys[i], b = keys[q];
        if ((a < b) == desc) {
          keys[i] = b;
          keys[q] = a;
        }
      }
      __syncthreads();
    }
    for (int base = 0; base < n_pow2; base += tile_n) {
      for (int i = threadIdx.x; i < tile_n; i += blockDim.x) lds[i] = keys[base + i];
      __syncthreads();
      bitonic_local(lds, tile_n, base, k, tile_n >> 1);
      for (int i = threadIdx.x; i < tile_n; i += blockDim.x) keys[base + i] = lds[i];
      __syncthreads();
    }
  }
}

// ---- workspace layout of the core ------------------------------------------------------------
struct CoreWs {
  u64* keys;        // [B, key_cap]   (key_cap = power of two)
  int32_t* n_keys;  // [B] appended candidates (prepare kernels atomically increment)
  int32_t* nc;      // [B] candidates after sort/clamp
  float4* sbox;     // [B, cap] sorted boxes
  int32_t* scls;    // [B, cap] sorted classes
  int32_t* sidx;    // [B, cap] sorted candidate indices
  u64* mask;        // [B, cap, words]
  u64* kept;        // [B, words] window state: kept bits of every resolved chunk
  int32_t* st_count;  // [B] kept so far
  int32_t* st_done;   // [B] 1 once the image is finished (max_out reached or every candidate resolved)
  int key_cap, cap, words;
  size_t bytes;
};

inline CoreWs core_layout(void* base, int B, int n_keys_max, int max_candidates) {
  CoreWs w;
  int cap = n_keys_max;
  if (max_candidates > 0 && max_candidates < cap) cap = max_candidates;
  if (cap < 1) cap = 1;
  w.cap = cap;
  w.words = (cap + 63) / 64;
  w.key_cap = hd_next_pow2(n_keys_max < 1 ? 1 : n_keys_max);
  size_t off = 0;
  char* p = reinterpret_cast<char*>(base);
  auto take = [&](size_t bytes) {
    char* r = p ? p + off : nullptr;
    off += hd_align_up(bytes, 256);
    return r;
  };
  w.keys = reinterpret_cast<u64*>(take((size_t)B * w.key_cap * sizeof(u64)));
  w.n_keys = reinterpret_cast<int32_t*>(take((size_t)B * sizeof(int32_t)));
  w.nc = reinterpret_cast<int32_t*>(take((size_t)B * sizeof(int32_t)));
  w.sbox = reinterpret_cast<float4*>(take((size_t)B * cap * sizeof(float4)));
  w.scls = reinterpret_cast<int32_t*>(take((size_t)B * cap * sizeof(int32_t)));
  w.sidx = reinterpret_cast<int32_t*>(take((size_t)B * cap * sizeof(int32_t)));
  w.mask = reinterpret_cast<u64*>(take((size_t)B * cap * w.words * sizeof(u64)));
  w.kept = reinterpret_cast<u64*>(take((size_t)B * w.words * sizeof(u64)));
  w.st_count = reinterpret_cast<int32_t*>(take((size_t)B * sizeof(int32_t)));
  w.st_done = reinterpret_cast<int32_t*>(take((size_t)B * sizeof(int32_t)));
  w.bytes = off;
  return w;
}

// ---- step 1 ---------------------------------------------------------------------------------
// cand_box / cand_cls are indexed by candidate index (stride cand_stride per image).
static __global__ __launch_bounds__(kSortThreads) void nms_sort_gather_kernel(CoreWs w, const float4* __restrict__ cand_box,
                                                                        const int32_t* __restrict__ cand_cls,
                                                                        long long cand_stride) {
  extern __shared__ u64 s_tile[];  // min(key_cap, kSortTile) keys
  const int b = blockIdx.x;
  u64* keys = w.keys + (size_t)b * w.key_cap;
  int n = w.n_keys[b];
  if (n > w.key_cap) n = w.key_cap;
  int n_pow2 = 1;
  while (n_pow2 < n) n_pow2 <<= 1;
  for (int i = n + threadIdx.x; i < n_pow2; i += blockDim.x) keys[i] = 0ull;
  __syncthreads();
  block_sort_desc(keys, n_pow2, s_tile, w.key_cap < kSortTile ? w.key_cap : kSortTile);
  const int nc = n < w.cap ? n : w.cap;
  if (threadIdx.x == 0) {
    w.nc[b] = nc;
    w.st_count[b] = 0;  // window state of steps 2 and 3
    w.st_done[b] = 0;
  }
  for (int r = threadIdx.x; r < nc; r += blockDim.x) {
    const uint32_t idx = key_index(keys[r]);
    w.sbox[(size_t)b * w.cap + r] = cand_box[(long long)b * cand_stride + idx];
    w.scls[(size_t)b * w.cap + r] = cand_cls ? cand_cls[(long long)b * cand_stride + idx] : 0;
    w.sidx[(size_t)b * w.cap + r] = (int32_t)idx;
  }
}

// ---- step 2 ---------------------------------------------------------------------------------
// Persistent over the ACTUAL work: the number of candidates nc[b] is only known on the device, so the
// grid is (kMaskBlocks, B) regardless of the capacity and every wave walks the upper-triangle tile
// pairs p = wave, wave + n_waves, ... of its image (p -> (row tile, column tile) by triangular
// indexing).  An image with 17 candidates costs one wave one tile; a capacity-sized grid would launch
// hundreds of thousands of workgroups that only read nc and exit.
// Per pair: the tile's 64 row boxes (+ areas, classes) go through a wave-private LDS slice and are
// read back as wave-uniform broadcasts, four rows per iteration so the LDS latency of one row
// overlaps the tests of the others; the suppression test of one row against the wave's 64 column
// boxes is one v_cmp whose 64-bit result IS the mask word (the wavefront ballot).
constexpr int kMaskBlocks = 512;  // x 4 waves per image

static __global__ __launch_bounds__(256) void nms_mask_kernel(CoreWs w, float iou_thr, int ct_lo, int ct_hi) {
  __shared__ float4 s_row[4][64];
  __shared__ float s_rarea[4][64];
  __shared__ int32_t s_rcls[4][64];
  const int b = blockIdx.y;
  if (w.st_done[b]) return;  // an earlier window already kept max_out boxes (or resolved every candidate)
  const int nc = w.nc[b];
  const int T = (nc + 63) >> 6;
  const int hi = ct_hi < T ? ct_hi : T;
  if (ct_lo >= hi) return;
  // tile pairs (rt <= ct) with ct in [ct_lo, hi), enumerated column by column: q = ct (ct + 1) / 2 + rt
  const long long q0 = (long long)ct_lo * (ct_lo + 1) / 2, q1 = (long long)hi * (hi + 1) / 2;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float4* sbox = w.sbox + (size_t)b * w.cap;
  const int32_t* scls = w.scls + (size_t)b * w.cap;
  const long long n_waves = (long long)gridDim.x * 4;
  for (long long q = q0 + (long long)blockIdx.x * 4 + wv; q < q1; q += n_waves) {
    int ct = (int)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
    while ((long long)ct * (ct + 1) / 2 > q) --ct;
    while ((long long)(ct + 1) * (ct + 2) / 2 <= q) ++ct;
    const int rt = (int)(q - (long long)ct * (ct + 1) / 2);
    const int row0 = rt * 64;
    {
      const int r = row0 + lane;
      const float4 v = r < nc ? sbox[r] : make_float4(0.f, 0.f, 0.f, 0.f);
      s_row[wv][lane] = v;
      s_rarea[wv][lane] = hd_box_area(v);
      s_rcls[wv][lane] = r < nc ? scls[r] : -1;
    }
    __builtin_amdgcn_wave_barrier();  // the slice is private to this wave: no workgroup barrier needed
    const int col = ct * 64 + lane;
    const bool col_ok = col < nc;
    const float4 cb = col_ok ? sbox[col] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int32_t cc = col_ok ? scls[col] : -2;
    const float carea = hd_box_area(cb);
    const int rows = min(64, nc - row0);
    u64 mine = 0ull;
    auto test = [&](const float4 rb, const float ra, const int rc) -> bool {
      if (col_ok && rc == cc && hd_overlaps(rb, cb)) return hd_iou_full(rb, ra, cb, carea) > iou_thr;
      return false;
    };
    int r = 0;
    for (; r + 3 < rows; r += 4) {
      const float4 b0 = s_row[wv][r], b1 = s_row[wv][r + 1], b2 = s_row[wv][r + 2], b3 = s_row[wv][r + 3];
      const float a0 = s_rarea[wv][r], a1 = s_rarea[wv][r + 1], a2 = s_rarea[wv][r + 2], a3 = s_rarea[wv][r + 3];
      const int c0 = s_rcls[wv][r], c1 = s_rcls[wv][r + 1], c2 = s_rcls[wv][r + 2], c3 = s_rcls[wv][r + 3];
      u64 w0 = __ballot(test(b0, a0, c0)), w1 = __ballot(test(b1, a1, c1));
      u64 w2 = __ballot(test(b2, a2, c2)), w3 = __ballot(test(b3, a3, c3));
      if (ct == rt) {  // only later boxes can be suppressed by row r
        w0 &= ~((2ull << r) - 1ull);
        w1 &= ~((2ull << (r + 1)) - 1ull);
        w2 &= ~((2ull << (r + 2)) - 1ull);
        w3 &= ~((2ull << (r + 3)) - 1ull);
      }
      if (lane == r) mine = w0;
      if (lane == r + 1) mine = w1;
      if (lane == r + 2) mine = w2;
      if (lane == r + 3) mine = w3;
    }
    for (; r < rows; ++r) {
      u64 word = __ballot(test(s_row[wv][r], s_rarea[wv][r], s_rcls[wv][r]));
      if (ct == rt) word &= ~((2ull << r) - 1ull);
      if (lane == r) mine = word;
    }
    if (lane < rows) w.mask[((size_t)b * w.cap + row0 + lane) * w.words + ct] = mine;
    __builtin_amdgcn_wave_barrier();  // all reads of the slice are done before the next pair overwrites it
  }
}

// ---- step 3 ---------------------------------------------------------------------------------
// keep_out[b, 0:count] = candidate indices in kept order; tail = -1.
static __global__ __launch_bounds__(1024) void nms_scan_kernel(CoreWs w, int max_out, int32_t* __restrict__ keep_out,
                                                        int32_t* __restrict__ keep_count, int c_lo, int c_hi) {
  __shared__ u64 s_cur;
  __shared__ u64 s_kept;
  __shared__ int s_count;
  const int b = blockIdx.x;
  if (w.st_done[b]) return;
  const int nc = w.nc[b];
  const int words = (nc + 63) / 64;
  const int t = threadIdx.x;
  const u64* mask = w.mask + (size_t)b * w.cap * w.words;
  const int32_t* sidx = w.sidx + (size_t)b * w.cap;
  int32_t* keep = keep_out + (size_t)b * max_out;
  u64* st_kept = w.kept + (size_t)b * w.words;
  const int hi = c_hi < words ? c_hi : words;
  u64 removed = 0ull;
  int count = c_lo == 0 ? 0 : w.st_count[b];
  // OR the mask words of the rows `kept` of chunk c into this lane's removed-word (column tile t), 8 independent loads
  // at a time (a plain one-load-per-iteration loop waits out a full memory latency per kept box)
  auto fold = [&](int c, u64 kept) {
    const u64* col = mask + (size_t)c * 64 * w.words + t;
    while (kept) {
      int j[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        j[k] = kept ? __ffsll((long long)kept) - 1 : -1;
        kept &= kept - 1;  // 0 & anything stays 0
      }
      u64 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = j[k] >= 0 ? col[(size_t)j[k] * w.words] : 0ull;
#pragma unroll
      for (int k = 0; k < 8; ++k) removed |= v[k];
    }
  };
  if (c_lo > 0 && t >= c_lo && t < hi) {
    // columns new in this window: the rows kept in earlier windows have not been applied to them yet
    for (int c = 0; c < c_lo; ++c) fold(c, st_kept[c]);
  }
  int c = c_lo;
  for (; c < hi && count < max_out; ++c) {
    if (t == c) s_cur = removed;
    __syncthreads();
    if (t < 64) {
      const int row = c * 64 + t;
      const u64 diag = row < nc ? mask[(size_t)row * w.words + c] : 0ull;
      // the whole resolve runs on scalar registers: the chunk's removed-word, the running kept mask and
      // the diagonal words fetched with v_readlane (a 64-bit __shfl would be an LDS round trip per step)
      const uint32_t dlo = (uint32_t)diag, dhi = (uint32_t)(diag >> 32);
      const u64 cur_v = s_cur;
      u64 cur = ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(cur_v >> 32)) << 32) |
                (u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)cur_v);
      const int rows = nc - c * 64;
      if (rows < 64) cur |= ~0ull << rows;  // rows past the candidates count as removed
      u64 kept = 0ull;
      u64 avail = ~cur;
      while (avail) {
        const int j = __builtin_amdgcn_readfirstlane(__ffsll((long long)avail) - 1);
        kept |= 1ull << j;
        const u64 dj = ((u64)(uint32_t)__builtin_amdgcn_readlane((int)dhi, j) << 32) |
                       (u64)(uint32_t)__builtin_amdgcn_readlane((int)dlo, j);
        cur |= dj;
        avail = ~cur & ~((2ull << j) - 1ull);
      }
      if ((kept >> t) & 1ull) {
        const int pos = count + __popcll(kept & ((1ull << t) - 1ull));
        if (pos < max_out) keep[pos] = sidx[row];
      }
      if (t == 0) {
        s_kept = kept;
        s_count = count + __popcll(kept);
        st_kept[c] = kept;
      }
    }
    __syncthreads();
    const u64 kept = s_kept;
    count = s_count;
    if (count >= max_out) {
      ++c;
      break;
    }
    if (t > c && t < hi) fold(c, kept);  // columns beyond this window are brought up to date when their window opens
    __syncthreads();  // s_cur / s_kept are rewritten next iteration
  }
  if (count >= max_out || c >= words) {
    // finished: max_out reached, or every candidate resolved
    if (count > max_out) count = max_out;
    if (t == 0) {
      keep_count[b] = count;
      w.st_done[b] = 1;
    }
    for (int i = count + t; i < max_out; i += blockDim.x) keep[i] = -1;
  } else if (t == 0) {
    w.st_count[b] = count;
  }
}

// Runs steps 1-3.  The caller has zeroed w.n_keys before its prepare kernel ran.
inline hd_status run_core(const CoreWs& w, int B, const float4* cand_box, const int32_t* cand_cls,
                          long long cand_stride, float iou_thr, int max_out, int32_t* keep_out,
                          int32_t* keep_count, hipStream_t s) {
  const int tile = w.key_cap < kSortTile ? w.key_cap : kSortTile;
  const size_t sort_lds = (size_t)tile * sizeof(u64);
  // > 64 KiB of dynamic LDS needs the opt-in once per DEVICE (the attribute lives with the device's code
  // object).  One flag per device ordinal; a race between two threads only repeats an idempotent host call.
  static std::atomic<unsigned char> attr_set[kMaxDevices];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return HD_ERR_NO_DEVICE;
  const bool tracked = dev >= 0 && dev < kMaxDevices;
  if (!tracked || !attr_set[dev].load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sort_gather_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kSortTile * (int)sizeof(u64)) != hipSuccess)
      return HD_ERR_LAUNCH;
    if (tracked) attr_set[dev].store(1, std::memory_order_release);
  }
  hipLaunchKernelGGL(nms_sort_gather_kernel, dim3(B), dim3(kSortThreads), sort_lds, s, w, cand_box, cand_cls,
                     cand_stride);
  HD_RETURN_IF_LAUNCH_FAILED();
  // one lane per 64-candidate mask word; fewer waves make the three barriers per chunk cheaper
  int scan_threads = ((w.words + 63) / 64) * 64;
  if (scan_threads < 64) scan_threads = 64;
  if (scan_threads > 1024) scan_threads = 1024;
  // windows over the sorted candidates, in 64-candidate chunks: [0,16) [16,64) [64,256) [256,1024); how many of them an
  // image needs is decided on the device (st_done), the host only bounds them by the workspace capacity
  const int window_end[4] = {16, 64, 256, 1024};
  int lo = 0;
  for (int k = 0; k < 4 && lo < w.words; ++k) {
    const int hi = window_end[k];
    // the first window's tile pairs (<= 136) fit a few workgroups per image; later ones use the persistent grid
    const long long pairs = (long long)(hi < w.words ? hi : w.words) * ((hi < w.words ? hi : w.words) + 1) / 2 - (long long)lo * (lo + 1) / 2;
    long long blocks = (pairs + 3) / 4;
    if (blocks > kMaskBlocks) blocks = kMaskBlocks;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(nms_mask_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, s, w, iou_thr, lo, hi);
    HD_RETURN_IF_LAUNCH_FAILED();
    hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(scan_threads), 0, s, w, max_out, keep_out, keep_count, lo, hi);
    HD_RETURN_IF_LAUNCH_FAILED();
    lo = hi;
  }
  return HD_OK;
}

}  // namespace hdnms
