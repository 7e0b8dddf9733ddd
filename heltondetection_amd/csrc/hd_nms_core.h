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
//   2. nms_mask_kernel          64x64 tiles of the upper triangle.  A wave holds 64 column boxes
//      (one per lane), the 64 row boxes are staged in LDS and broadcast; the suppression test of
//      one row against the 64 columns is a single v_cmp whose 64-bit result IS the mask word
//      (the wavefront ballot).  Non-overlapping or other-class pairs exit after ~8 VALU ops.
//   3. nms_scan_kernel          one workgroup per image; lane t owns word t of the removed-set.
//      Per 64-box chunk wave 0 resolves the chunk against its diagonal word with scalar
//      bit-ops, then all lanes OR the kept rows' words (independent loads).  Stops at max_out.
//
// The result is bit-identical to the serial algorithm in oracle/hd_oracle.c: nms_one().
#pragma once
#include "hd_common.h"

namespace hdnms {

constexpr int kSortThreads = 1024;
constexpr int kSortTile = 4096;   // keys per LDS tile (32 KiB)
constexpr int kMaxCand = 65536;   // 1024 lanes x one 64-bit removed-word each

typedef unsigned long long u64;

__device__ __forceinline__ u64 make_key(float score, uint32_t index) {
  return ((u64)hd_f2key(score) << 32) | (u64)(0xFFFFFFFFu - index);
}
__device__ __forceinline__ uint32_t key_index(u64 k) { return 0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull); }
__device__ __forceinline__ float key_score(u64 k) { return hd_key2f((uint32_t)(k >> 32)); }

// compare-exchange steps j = j_from, j_from/2, ..., 1 of bitonic stage k on an LDS tile whose
// first element has global index `base`; final order is DESCENDING.
__device__ __forceinline__ void bitonic_local(u64* lds, int tile_n, int base, int k, int j_from) {
  for (int j = j_from; j > 0; j >>= 1) {
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
  }
}

// Sorts keys[0..n_pow2) (global memory) descending with the calling workgroup.
// lds must hold kSortTile keys.  Every thread of the workgroup must call this.
__device__ __forceinline__ void block_sort_desc(u64* keys, int n_pow2, u64* lds) {
  const int tile_n = n_pow2 < kSortTile ? n_pow2 : kSortTile;
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
  w.bytes = off;
  return w;
}

// ---- step 1 ---------------------------------------------------------------------------------
// cand_box / cand_cls are indexed by candidate index (stride cand_stride per image).
static __global__ __launch_bounds__(kSortThreads) void nms_sort_gather_kernel(CoreWs w, const float4* __restrict__ cand_box,
                                                                        const int32_t* __restrict__ cand_cls,
                                                                        long long cand_stride) {
  __shared__ u64 s_tile[kSortTile];
  const int b = blockIdx.x;
  u64* keys = w.keys + (size_t)b * w.key_cap;
  int n = w.n_keys[b];
  if (n > w.key_cap) n = w.key_cap;
  int n_pow2 = 1;
  while (n_pow2 < n) n_pow2 <<= 1;
  for (int i = n + threadIdx.x; i < n_pow2; i += blockDim.x) keys[i] = 0ull;
  __syncthreads();
  block_sort_desc(keys, n_pow2, s_tile);
  const int nc = n < w.cap ? n : w.cap;
  if (threadIdx.x == 0) w.nc[b] = nc;
  for (int r = threadIdx.x; r < nc; r += blockDim.x) {
    const uint32_t idx = key_index(keys[r]);
    w.sbox[(size_t)b * w.cap + r] = cand_box[(long long)b * cand_stride + idx];
    w.scls[(size_t)b * w.cap + r] = cand_cls ? cand_cls[(long long)b * cand_stride + idx] : 0;
    w.sidx[(size_t)b * w.cap + r] = (int32_t)idx;
  }
}

// ---- step 2 ---------------------------------------------------------------------------------
// grid: (ceil(words/4), words, B); block 256 = 4 waves, wave v handles column tile 4*bx + v.
static __global__ __launch_bounds__(256) void nms_mask_kernel(CoreWs w, float iou_thr) {
  __shared__ float4 s_row[64];
  __shared__ int32_t s_rcls[64];
  const int b = blockIdx.z;
  const int nc = w.nc[b];
  const int rt = blockIdx.y;
  const int row0 = rt * 64;
  if (row0 >= nc) return;  // whole block exits together
  const int lane = threadIdx.x & 63;
  const int ct = blockIdx.x * 4 + (threadIdx.x >> 6);
  const float4* sbox = w.sbox + (size_t)b * w.cap;
  const int32_t* scls = w.scls + (size_t)b * w.cap;
  if (threadIdx.x < 64) {
    const int r = row0 + threadIdx.x;
    s_row[threadIdx.x] = r < nc ? sbox[r] : make_float4(0.f, 0.f, 0.f, 0.f);
    s_rcls[threadIdx.x] = r < nc ? scls[r] : -1;
  }
  __syncthreads();
  if (ct < rt || ct * 64 >= nc) return;  // lower triangle / beyond the candidates (wave-uniform)
  const int col = ct * 64 + lane;
  const bool col_ok = col < nc;
  const float4 cb = col_ok ? sbox[col] : make_float4(0.f, 0.f, 0.f, 0.f);
  const int32_t cc = col_ok ? scls[col] : -2;
  const float carea = hd_box_area(cb);
  const int rows = min(64, nc - row0);
  u64 mine = 0ull;
  for (int r = 0; r < rows; ++r) {
    const float4 rb = s_row[r];
    bool sup = false;
    if (col_ok && s_rcls[r] == cc && hd_overlaps(rb, cb))
      sup = hd_iou_full(rb, hd_box_area(rb), cb, carea) > iou_thr;
    u64 word = __ballot(sup);
    if (ct == rt) word &= ~((2ull << r) - 1ull);  // only later boxes can be suppressed by row r
    if (lane == r) mine = word;
  }
  if (lane < rows) w.mask[((size_t)b * w.cap + row0 + lane) * w.words + ct] = mine;
}

// ---- step 3 ---------------------------------------------------------------------------------
// keep_out[b, 0:count] = candidate indices in kept order; tail = -1.
static __global__ __launch_bounds__(1024) void nms_scan_kernel(CoreWs w, int max_out, int32_t* __restrict__ keep_out,
                                                        int32_t* __restrict__ keep_count) {
  __shared__ u64 s_cur;
  __shared__ u64 s_kept;
  __shared__ int s_count;
  const int b = blockIdx.x;
  const int nc = w.nc[b];
  const int words = (nc + 63) / 64;
  const int t = threadIdx.x;
  const u64* mask = w.mask + (size_t)b * w.cap * w.words;
  const int32_t* sidx = w.sidx + (size_t)b * w.cap;
  int32_t* keep = keep_out + (size_t)b * max_out;
  u64 removed = 0ull;
  int count = 0;
  for (int c = 0; c < words && count < max_out; ++c) {
    if (t == c) s_cur = removed;
    __syncthreads();
    if (t < 64) {
      const int row = c * 64 + t;
      const u64 diag = row < nc ? mask[(size_t)row * w.words + c] : 0ull;
      u64 cur = s_cur;
      const int rows = nc - c * 64;
      if (rows < 64) cur |= ~0ull << rows;  // rows past the candidates count as removed
      u64 kept = 0ull;
      u64 avail = ~cur;
      while (avail) {
        const int j = __ffsll((long long)avail) - 1;
        kept |= 1ull << j;
        cur |= __shfl(diag, j, 64);
        avail = ~cur & ~((2ull << j) - 1ull);
      }
      if ((kept >> t) & 1ull) {
        const int pos = count + __popcll(kept & ((1ull << t) - 1ull));
        if (pos < max_out) keep[pos] = sidx[row];
      }
      if (t == 0) {
        s_kept = kept;
        s_count = count + __popcll(kept);
      }
    }
    __syncthreads();
    u64 kept = s_kept;
    count = s_count;
    if (count >= max_out) break;
    if (t > c && t < words) {
      while (kept) {
        const int j = __ffsll((long long)kept) - 1;
        kept &= kept - 1;
        removed |= mask[(size_t)(c * 64 + j) * w.words + t];
      }
    }
    __syncthreads();  // s_cur / s_kept are rewritten next iteration
  }
  if (count > max_out) count = max_out;
  if (t == 0) keep_count[b] = count;
  for (int i = count + t; i < max_out; i += blockDim.x) keep[i] = -1;
}

// Runs steps 1-3.  The caller has zeroed w.n_keys before its prepare kernel ran.
inline hd_status run_core(const CoreWs& w, int B, const float4* cand_box, const int32_t* cand_cls,
                          long long cand_stride, float iou_thr, int max_out, int32_t* keep_out,
                          int32_t* keep_count, hipStream_t s) {
  hipLaunchKernelGGL(nms_sort_gather_kernel, dim3(B), dim3(kSortThreads), 0, s, w, cand_box, cand_cls, cand_stride);
  HD_RETURN_IF_LAUNCH_FAILED();
  dim3 mgrid((unsigned)((w.words + 3) / 4), (unsigned)w.words, (unsigned)B);
  hipLaunchKernelGGL(nms_mask_kernel, mgrid, dim3(256), 0, s, w, iou_thr);
  HD_RETURN_IF_LAUNCH_FAILED();
  hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(1024), 0, s, w, max_out, keep_out, keep_count);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

}  // namespace hdnms
