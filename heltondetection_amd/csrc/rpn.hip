// rpn.hip — RPN proposal generation (reference README.md:8; SURVEY.md §8a "RPN proposal top-K + NMS").
//
//   rpn_topk_kernel   one 1024-lane workgroup per (level, image): exact top-k of the objectness
//                     logits by MSB-first radix select on the 64-bit key (logit bits, ~index) —
//                     8-bit digits, LDS histogram per pass, early exit as soon as the remaining
//                     bucket is taken whole (4 passes + 1 select pass for distinct logits).  The
//                     selected anchors are decoded, clipped, size-filtered and appended as NMS
//                     candidates; nothing of the other N-k anchors is ever written.
//   engine            hd_nms_core.h with class id = pyramid level (per-level NMS), max_out =
//                     post_nms_topk.
//   rpn_emit_kernel   gathers the kept boxes, sigmoid(logit) scores, zero padding.
//
// Ties: keys are (logit, lower anchor index first) everywhere, which equals the oracle's
// "per-level rank order, then candidate order" because levels are concatenated in index order.
#include "hd_ciou.h"
#include "hd_nms_core.h"

using namespace hdnms;

namespace {

struct RpnLevels {
  int n_levels;
  int off[HD_MAX_LEVELS + 1];
};

// Large levels are cut into chunks (round 3): one workgroup walking a 130 000-logit level eight times was 349 us on average;
// a chunk's exact top-k is found by its own workgroup (rpn_chunk_topk_kernel) and the level's workgroup then selects among
// the chunks' survivors only.  Any element of the level's top-k is in its chunk's top-k, and keys are unique, so the kept
// set is the same bit for bit.
constexpr int kMaxChunks = 64;
struct RpnChunks {
  int n;                      // chunks in all (0: no level is chunked)
  int level[kMaxChunks];      // level of chunk c
  int start[kMaxChunks];      // first anchor index (global) of chunk c
  int len[kMaxChunks];
  int first[HD_MAX_LEVELS];   // first chunk of level l, or -1 when the level is walked directly
  int count[HD_MAX_LEVELS];   // chunks of level l
};

// MSB-first radix select on 64-bit keys: the threshold key of the k largest among n candidates, `load(i, &key)` returning
// whether candidate i exists.  8-bit digits, LDS histogram per pass, early exit as soon as the remaining bucket is taken
// whole.  Called by every thread of the workgroup; s_hist[256], s_prefix, s_need, s_done are the caller's LDS.
template <class Load>
__device__ __forceinline__ u64 radix_select_threshold(Load load, int n, int k, int* s_hist, u64* s_prefix, int* s_need,
                                                      int* s_done) {
  u64 prefix = 0ull, pmask = 0ull;
  int need = k;
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 56 - 8 * pass;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    // Logit exponents cluster, so most lanes of a wave hit the same bin and a plain LDS atomic would serialise 64-fold:
    // 8 ballots find the lanes with an equal digit and the lowest such lane adds the whole group's count with one atomic.
    constexpr int U = 8;
    for (int i0 = 0; i0 < n; i0 += blockDim.x * U) {
      u64 key[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {  // 8 independent loads in flight per lane
        const int i = i0 + u * blockDim.x + threadIdx.x;
        key[u] = 0ull;
        ok[u] = i < n && load(i, &key[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool in = ok[u] && ((key[u] & pmask) == prefix);
        const int digit = (int)((key[u] >> shift) & 255ull);
        u64 peers = __ballot(in);
        if (peers == 0ull) continue;  // wave-uniform
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
          const bool one = (digit >> bit) & 1;
          const u64 bal = __ballot(in && one);
          peers &= one ? bal : ~bal;
        }
        if (in && (__ffsll((long long)peers) - 1) == (int)(threadIdx.x & 63)) atomicAdd(&s_hist[digit], __popcll(peers));
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int above = 0, d = 255;
      for (; d > 0; --d) {
        if (above + s_hist[d] >= need) break;
        above += s_hist[d];
      }
      *s_need = need - above;
      *s_prefix = prefix | ((u64)d << shift);
      *s_done = (s_hist[d] == need - above) ? 1 : 0;
    }
    __syncthreads();
    need = *s_need;
    prefix = *s_prefix;
    pmask |= 255ull << shift;
    const int done = *s_done;
    __syncthreads();
    if (done) break;
  }
  return prefix;
}

// One workgroup per (chunk, image): the anchor indices of the chunk's min(k, len) best logits -> sel_idx[b][chunk][0..k),
// their number -> sel_cnt[b][chunk] (order irrelevant: the level's workgroup selects among them by key).
__global__ __launch_bounds__(1024) void rpn_chunk_topk_kernel(const float* __restrict__ logits, int NA, RpnChunks ch,
                                                              int pre_nms_topk, int32_t* __restrict__ sel_idx,
                                                              int32_t* __restrict__ sel_cnt) {
  __shared__ int s_hist[256];
  __shared__ u64 s_prefix;
  __shared__ int s_need, s_done, s_cnt;
  const int c = blockIdx.x, b = blockIdx.y;
  const int start = ch.start[c], n = ch.len[c];
  const int k = n < pre_nms_topk ? n : pre_nms_topk;
  const float* lg = logits + (size_t)b * NA;
  u64 thr = 0ull;
  if (k < n) {
    auto load = [&](int i, u64* key) {
      *key = make_key(lg[start + i], (uint32_t)(start + i));
      return true;
    };
    thr = radix_select_threshold(load, n, k, s_hist, &s_prefix, &s_need, &s_done);
  }
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  int32_t* out = sel_idx + ((size_t)b * kMaxChunks + c) * pre_nms_topk;
  const int lane = threadIdx.x & 63;
  const int n_round = (n + (int)blockDim.x - 1) / (int)blockDim.x * (int)blockDim.x;
  for (int i = threadIdx.x; i < n_round; i += blockDim.x) {
    const bool take = i < n && make_key(lg[start + i], (uint32_t)(start + i)) >= thr;
    const u64 m = __ballot(take);
    if (m != 0ull) {
      int base = 0;
      if (lane == (int)(__ffsll((long long)m) - 1)) base = atomicAdd(&s_cnt, __popcll(m));
      base = __shfl(base, __ffsll((long long)m) - 1, 64);
      if (take) {
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        if (pos < pre_nms_topk) out[pos] = start + i;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) sel_cnt[(size_t)b * kMaxChunks + c] = min(s_cnt, pre_nms_topk);
}

__global__ __launch_bounds__(1024) void rpn_topk_kernel(const float* __restrict__ logits,
                                                        const float4* __restrict__ deltas,
                                                        const float4* __restrict__ anchors, int NA, RpnLevels lv,
                                                        int pre_nms_topk, float min_size, float img_h, float img_w,
                                                        CoreWs w, float4* __restrict__ cand_box,
                                                        int32_t* __restrict__ cand_cls, int stage_cap, RpnChunks ch,
                                                        const int32_t* __restrict__ sel_idx,
                                                        const int32_t* __restrict__ sel_cnt) {
  __shared__ int s_hist[256];
  __shared__ u64 s_prefix;
  __shared__ int s_need;
  __shared__ int s_done;
  const int l = blockIdx.x, b = blockIdx.y;
  const int off = lv.off[l];
  const int n_level = lv.off[l + 1] - lv.off[l];
  const int k = n_level < pre_nms_topk ? n_level : pre_nms_topk;
  const float* lg = logits + (size_t)b * NA;
  // a chunked level: the candidates are the survivors of its chunks, slot i = (chunk i / pre, position i % pre)
  const bool chunked = ch.n > 0 && ch.first[l] >= 0;
  const int32_t* my_sel = chunked ? sel_idx + ((size_t)b * kMaxChunks + ch.first[l]) * pre_nms_topk : nullptr;
  const int32_t* my_cnt = chunked ? sel_cnt + (size_t)b * kMaxChunks + ch.first[l] : nullptr;
  const int n = chunked ? ch.count[l] * pre_nms_topk : n_level;
  auto anchor_of = [&](int i) -> int {   // global anchor index of candidate i, or -1
    if (!chunked) return off + i;
    const int c = i / pre_nms_topk, j = i - c * pre_nms_topk;
    return j < my_cnt[c] ? my_sel[i] : -1;
  };
  u64 thr = 0ull;  // select keys >= thr
  if (chunked) {
    auto load = [&](int i, u64* key) {
      const int gi = anchor_of(i);
      if (gi < 0) return false;
      *key = make_key(lg[gi], (uint32_t)gi);
      return true;
    };
    thr = radix_select_threshold(load, n, k, s_hist, &s_prefix, &s_need, &s_done);
  } else if (k < n) {
    auto load = [&](int i, u64* key) {
      *key = make_key(lg[off + i], (uint32_t)(off + i));
      return true;
    };
    thr = radix_select_threshold(load, n, k, s_hist, &s_prefix, &s_need, &s_done);
  }
  u64* keys = w.keys + (size_t)b * w.key_cap;
  // Selected keys are staged in LDS (positions from an LDS counter, one ds_add per wave via a ballot) and the
  // image's global counter is bumped ONCE per workgroup: n_keys[] of all images share a cache line, and one
  // returning atomic per selected anchor (up to pre_nms_topk x levels x images of them) serialised there.
  // stage_cap == 0 (pre_nms_topk too large for LDS): the old per-anchor reservation.
  extern __shared__ u64 s_stage[];
  __shared__ int s_cnt, s_base;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int n_round = (n + (int)blockDim.x - 1) / (int)blockDim.x * (int)blockDim.x;  // whole waves reach the ballot
#pragma unroll 2
  for (int i = threadIdx.x; i < n_round; i += blockDim.x) {
    const int gi = i < n ? anchor_of(i) : -1;
    bool take = false;
    u64 key = 0ull;
    if (gi >= 0) {
      const float v = lg[gi];
      key = make_key(v, (uint32_t)gi);
      if (key >= thr) {
        const float4 bx = hd_delta_decode(anchors[gi], deltas[(size_t)b * NA + gi], img_h, img_w);
        if ((bx.z - bx.x >= min_size) && (bx.w - bx.y >= min_size)) {
          cand_box[(size_t)b * NA + gi] = bx;
          cand_cls[(size_t)b * NA + gi] = l;
          take = true;
        }
      }
    }
    if (stage_cap > 0) {
      const u64 m = __ballot(take);
      if (m != 0ull) {
        int base = 0;
        if (lane == (int)(__ffsll((long long)m) - 1)) base = atomicAdd(&s_cnt, __popcll(m));
        base = __shfl(base, __ffsll((long long)m) - 1, 64);
        if (take) {
          const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
          if (pos < stage_cap) s_stage[pos] = key;
        }
      }
    } else if (take) {
      const int pos = atomicAdd(&w.n_keys[b], 1);
      if (pos < w.key_cap) keys[pos] = key;
    }
  }
  if (stage_cap > 0) {
    __syncthreads();
    const int cnt = min(s_cnt, stage_cap);
    if (threadIdx.x == 0) s_base = atomicAdd(&w.n_keys[b], cnt);
    __syncthreads();
    const int base = s_base;
    for (int i = threadIdx.x; i < cnt; i += blockDim.x)
      if (base + i < w.key_cap) keys[base + i] = s_stage[i];
  }
}

__global__ __launch_bounds__(256) void rpn_emit_kernel(const int32_t* __restrict__ keep,
                                                       const int32_t* __restrict__ count, int NA, int post,
                                                       const float4* __restrict__ cand_box,
                                                       const float* __restrict__ logits, float4* __restrict__ rois,
                                                       float* __restrict__ roi_scores,
                                                       int32_t* __restrict__ roi_index /* may be null */) {
  const int b = blockIdx.y;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < post; k += gridDim.x * blockDim.x) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    float sc = 0.0f;
    int gi = -1;
    if (k < count[b]) {
      gi = keep[(size_t)b * post + k];
      r = cand_box[(size_t)b * NA + gi];
      sc = hd_sigmoid(logits[(size_t)b * NA + gi]);
    }
    rois[(size_t)b * post + k] = r;
    roi_scores[(size_t)b * post + k] = sc;
    if (roi_index) roi_index[(size_t)b * post + k] = gi;
  }
}

struct RpnWs {
  float4* cand_box;
  int32_t* cand_cls;
  int32_t* keep;
  int32_t* sel_idx;   // [B][kMaxChunks][pre]
  int32_t* sel_cnt;   // [B][kMaxChunks]
  size_t core_off, bytes;
};

constexpr int kChunkMin = 16384;  // levels up to this size are walked directly by their one workgroup

inline RpnWs rpn_layout(void* base, int B, int NA, int post, int pre) {
  RpnWs r;
  size_t off = 0;
  char* p = reinterpret_cast<char*>(base);
  auto take = [&](size_t bytes) {
    char* q = p ? p + off : nullptr;
    off += hd_align_up(bytes, 256);
    return q;
  };
  r.cand_box = reinterpret_cast<float4*>(take((size_t)B * NA * sizeof(float4)));
  r.cand_cls = reinterpret_cast<int32_t*>(take((size_t)B * NA * sizeof(int32_t)));
  r.keep = reinterpret_cast<int32_t*>(take((size_t)B * (post > 0 ? post : 1) * sizeof(int32_t)));
  const bool may_chunk = NA > kChunkMin;
  r.sel_idx = reinterpret_cast<int32_t*>(take(may_chunk ? (size_t)B * kMaxChunks * pre * sizeof(int32_t) : 0));
  r.sel_cnt = reinterpret_cast<int32_t*>(take(may_chunk ? (size_t)B * kMaxChunks * sizeof(int32_t) : 0));
  r.core_off = off;
  r.bytes = off;
  return r;
}

// upper bound of NMS candidates per image: every level contributes at most pre_nms_topk; the
// per-level sizes are not part of the workspace query, so bound by min(NA, n_levels*pre).
inline int cand_bound(int NA, int n_levels, int pre) {
  long long c = (long long)n_levels * pre;
  return (int)(c < NA ? c : NA);
}

}  // namespace

extern "C" size_t hd_rpn_proposals_workspace_bytes(int B, int NA, int n_levels, int pre_nms_topk,
                                                   int post_nms_topk) {
  if (B <= 0 || NA <= 0 || n_levels <= 0 || pre_nms_topk <= 0 || post_nms_topk <= 0) return 0;
  const int nk = cand_bound(NA, n_levels, pre_nms_topk);
  return rpn_layout(nullptr, B, NA, post_nms_topk, pre_nms_topk).bytes + core_layout(nullptr, B, nk, 0).bytes;
}

extern "C" hd_status hd_rpn_proposals(const float* logits, const float* deltas, const float* anchors, int B,
                                      int NA, int n_levels, const int* level_sizes, int pre_nms_topk,
                                      int post_nms_topk, float iou_thr, float min_size, float img_h, float img_w,
                                      float* rois, float* roi_scores, int32_t* roi_index, int32_t* count,
                                      void* workspace, size_t workspace_bytes, hd_stream stream) {
  if (B <= 0 || NA <= 0 || n_levels <= 0 || n_levels > HD_MAX_LEVELS || pre_nms_topk <= 0 || post_nms_topk <= 0 ||
      B > 65535)
    return HD_ERR_SHAPE;
  if (!logits || !deltas || !anchors || !level_sizes || !rois || !roi_scores || !count) return HD_ERR_NULL;
  if (!hd_aligned16(deltas) || !hd_aligned16(anchors) || !hd_aligned16(rois)) return HD_ERR_ALIGN;
  RpnLevels lv;
  lv.n_levels = n_levels;
  int off = 0;
  for (int l = 0; l < n_levels; ++l) {
    if (level_sizes[l] <= 0) return HD_ERR_SHAPE;
    lv.off[l] = off;
    off += level_sizes[l];
  }
  for (int l = n_levels; l <= HD_MAX_LEVELS; ++l) lv.off[l] = off;
  if (off != NA) return HD_ERR_SHAPE;
  const int nk = cand_bound(NA, n_levels, pre_nms_topk);
  if (!workspace) return HD_ERR_WORKSPACE;
  if (!hd_aligned16(workspace)) return HD_ERR_ALIGN;
  RpnWs r = rpn_layout(workspace, B, NA, post_nms_topk, pre_nms_topk);
  CoreWs w = core_layout(reinterpret_cast<char*>(workspace) + r.core_off, B, nk, 0);
  if (w.cap > kMaxCand) return HD_ERR_SHAPE;
  if (workspace_bytes < r.bytes + w.bytes) return HD_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  if (hd_fill32(w.n_keys, 0u, (size_t)B, s) != HD_OK) return HD_ERR_LAUNCH;
  // LDS staging of the selected keys: at most pre_nms_topk per (level, image); 6144 keys = 48 KiB
  const int stage_cap = pre_nms_topk <= 6144 ? pre_nms_topk : 0;
  // chunk the levels that are much larger than what they keep: chunk length >= kChunkMin, <= kMaxChunks chunks in all
  RpnChunks ch;
  ch.n = 0;
  for (int l = 0; l < HD_MAX_LEVELS; ++l) {
    ch.first[l] = -1;
    ch.count[l] = 0;
  }
  {
    int clen = kChunkMin;
    while ((long long)(NA + clen - 1) / clen + n_levels > kMaxChunks) clen *= 2;
    for (int l = 0; l < n_levels; ++l) {
      const int n = level_sizes[l];
      if (n < 2 * clen || n <= 4 * pre_nms_topk) continue;  // small, or barely reduced by a pre-selection
      const int cnt = (n + clen - 1) / clen;
      ch.first[l] = ch.n;
      ch.count[l] = cnt;
      for (int c = 0; c < cnt; ++c) {
        ch.level[ch.n] = l;
        ch.start[ch.n] = lv.off[l] + c * clen;
        ch.len[ch.n] = (c + 1 < cnt) ? clen : n - c * clen;
        ++ch.n;
      }
    }
  }
  if (ch.n > 0) {
    hipLaunchKernelGGL(rpn_chunk_topk_kernel, dim3(ch.n, B), dim3(1024), 0, s, logits, NA, ch, pre_nms_topk, r.sel_idx, r.sel_cnt);
    HD_RETURN_IF_LAUNCH_FAILED();
  }
  hipLaunchKernelGGL(rpn_topk_kernel, dim3(n_levels, B), dim3(1024), (size_t)stage_cap * sizeof(u64), s, logits,
                     reinterpret_cast<const float4*>(deltas), reinterpret_cast<const float4*>(anchors), NA, lv,
                     pre_nms_topk, min_size, img_h, img_w, w, r.cand_box, r.cand_cls, stage_cap, ch,
                     (const int32_t*)r.sel_idx, (const int32_t*)r.sel_cnt);
  HD_RETURN_IF_LAUNCH_FAILED();
  hd_status st = run_core(w, B, r.cand_box, r.cand_cls, (long long)NA, iou_thr, post_nms_topk, r.keep, count, s);
  if (st != HD_OK) return st;
  hipLaunchKernelGGL(rpn_emit_kernel, dim3((post_nms_topk + 255) / 256, B), dim3(256), 0, s, r.keep, count, NA,
                     post_nms_topk, r.cand_box, logits, reinterpret_cast<float4*>(rois), roi_scores, roi_index);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}
