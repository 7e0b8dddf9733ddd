// nms.hip — hd_nms_batched and hd_yolo_nms on top of the shared engine (hd_nms_core.h).
// SURVEY.md §8a rows "RPN proposal top-K + NMS" (class-agnostic / per-level) and "Per-class NMS".
#include <algorithm>

#include "hd_nms_core.h"

using namespace hdnms;

namespace {

// ---- hd_nms_batched: keys straight from the caller's scores ---------------------------------
__global__ __launch_bounds__(256) void nms_keys_kernel(const float* __restrict__ scores,
                                                       const int32_t* __restrict__ n_valid, int N, float score_thr,
                                                       CoreWs w) {
  const int b = blockIdx.y;
  const int nv = n_valid ? min(n_valid[b], N) : N;
  u64* keys = w.keys + (size_t)b * w.key_cap;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += gridDim.x * blockDim.x) {
    const float s = scores[(size_t)b * N + i];
    if (s > score_thr) {  // NaN fails
      const int pos = atomicAdd(&w.n_keys[b], 1);
      keys[pos] = make_key(s, (uint32_t)i);
    }
  }
}

// ---- hd_yolo_nms ----------------------------------------------------------------------------
// A wave takes 64 consecutive decoded rows: lane i tests the objectness of row base+i (64 loads in
// flight), one ballot finds the rows that pass, and for each of those all 64 lanes scan the class
// scores together (coalesced) and arg-max across lanes with ties -> lowest class.  Rows failing the
// objectness gate — almost all of them on a real detector — cost one 4-byte load and nothing else.
__global__ __launch_bounds__(256) void yolo_cand_kernel(const float* __restrict__ dec, int n, int no, float conf_thr,
                                                        CoreWs w, float4* __restrict__ cand_box,
                                                        int32_t* __restrict__ cand_cls,
                                                        float* __restrict__ cand_conf) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * blockDim.x) >> 6;
  u64* keys = w.keys + (size_t)b * w.key_cap;
  const float* base = dec + (size_t)b * n * no;
  for (int r0 = wave * 64; r0 < n; r0 += n_waves * 64) {
    const int row = r0 + lane;
    const float my_obj = row < n ? base[(size_t)row * no + 4] : 0.0f;
    unsigned long long pass = __ballot(row < n && my_obj > conf_thr);
    // lane j ends up holding the result of row r0 + j; nothing on this serial part waits for an atomic
    float my_conf = -1.0f;
    int my_cls = 0;
    while (pass) {
      const int j = __ffsll((long long)pass) - 1;
      pass &= pass - 1;
      const int i = r0 + j;
      const float* r = base + (size_t)i * no;
      const float obj = __shfl(my_obj, j, 64);
      float best = -1.0f;
      int bc = 0;
      for (int c = 5 + lane; c < no; c += 64) {
        const float v = r[c];
        if (v > best) {
          best = v;
          bc = c - 5;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oc = __shfl_xor(bc, o, 64);
        if (ob > best || (ob == best && oc < bc)) {
          best = ob;
          bc = oc;
        }
      }
      if (lane == j) {
        my_conf = obj * best;
        my_cls = bc;
      }
    }
    // one reservation per 64-row chunk (was: one returning atomic per candidate; the per-image counters of a
    // batch share a cache line, and ~300k of those serialised into 3.8 ms on a dense batch), then every
    // passing lane writes its own record
    const bool take = my_conf > conf_thr;
    const unsigned long long m = __ballot(take);
    if (m != 0ull) {
      int pos0 = 0;
      if (lane == 0) pos0 = atomicAdd(&w.n_keys[b], __popcll(m));
      pos0 = __shfl(pos0, 0, 64);
      if (take) {
        const float* r = base + (size_t)row * no;
        const float cx = r[0], cy = r[1], bw = r[2], bh = r[3];
        cand_box[(size_t)b * n + row] = make_float4(cx - bw * 0.5f, cy - bh * 0.5f, cx + bw * 0.5f, cy + bh * 0.5f);
        cand_cls[(size_t)b * n + row] = my_cls;
        cand_conf[(size_t)b * n + row] = my_conf;
        keys[pos0 + __popcll(m & ((1ull << lane) - 1ull))] = make_key(my_conf, (uint32_t)row);
      }
    }
  }
}

__global__ __launch_bounds__(256) void yolo_det_kernel(const int32_t* __restrict__ keep,
                                                       const int32_t* __restrict__ det_count, int n, int max_det,
                                                       const float4* __restrict__ cand_box,
                                                       const int32_t* __restrict__ cand_cls,
                                                       const float* __restrict__ cand_conf, float* __restrict__ det) {
  const int b = blockIdx.y;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < max_det; k += gridDim.x * blockDim.x) {
    float* d = det + ((size_t)b * max_det + k) * 6;
    if (k < det_count[b]) {
      const int i = keep[(size_t)b * max_det + k];
      const float4 bx = cand_box[(size_t)b * n + i];
      d[0] = bx.x; d[1] = bx.y; d[2] = bx.z; d[3] = bx.w;
      d[4] = cand_conf[(size_t)b * n + i];
      d[5] = (float)cand_cls[(size_t)b * n + i];
    } else {
      d[0] = d[1] = d[2] = d[3] = d[4] = d[5] = 0.0f;
    }
  }
}

// ---- hd_yolo_detect -------------------------------------------------------------------------
// hd_yolo_decode + hd_yolo_nms without the decoded tensor: the candidate pass gates on the RAW objectness
// logit (sigmoid applied exactly as the decode kernel does) and decodes box + classes only for rows that
// pass, so inference never writes and re-reads B*n*no floats (137 MB at bs16 640^2 C=80).  Candidate index
// = the row index of the decoded layout (level, anchor, y, x), hence identical keys, ties and detections.
struct RawIndex {
  int first_row[HD_MAX_LEVELS + 1];
};

__global__ __launch_bounds__(256) void yolo_raw_cand_kernel(const hd_yolo_levels lv, const RawIndex ix, int n,
                                                            float conf_thr, CoreWs w, float4* __restrict__ cand_box,
                                                            int32_t* __restrict__ cand_cls,
                                                            float* __restrict__ cand_conf) {
  const int b = blockIdx.y, l = blockIdx.z;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int n_waves = (gridDim.x * blockDim.x) >> 6;
  const int na = lv.na, no = lv.no, H = lv.h[l], W = lv.w[l];
  const int rows_l = na * H * W;
  const long long so = lv.so[l];
  const float stride = lv.stride[l];
  u64* keys = w.keys + (size_t)b * w.key_cap;
  const float* base = lv.pred[l] + (long long)b * lv.sb[l];
  // rows are walked in MEMORY order of a channels-last head output, m = (y*W + x)*na + a, so the 64 objectness
  // logits of a wave sit 4*no bytes apart in one contiguous span; other layouts only change the address
  for (int m0 = wave * 64; m0 < rows_l; m0 += n_waves * 64) {
    const int m = m0 + lane;
    const bool ok = m < rows_l;
    const int a = ok ? m % na : 0;
    const int cell = ok ? m / na : 0;
    const int y = cell / W, x = cell - y * W;
    const float* r = base + (long long)a * lv.sa[l] + (long long)y * lv.sy[l] + (long long)x * lv.sx[l];
    const float my_obj = ok ? hd_sigmoid(r[4 * so]) : 0.0f;
    unsigned long long pass = __ballot(ok && my_obj > conf_thr);
    float my_conf = -1.0f;
    int my_cls = 0;
    while (pass) {
      const int j = __ffsll((long long)pass) - 1;
      pass &= pass - 1;
      // row j's address, rebuilt wave-uniformly
      const int mj = m0 + j;
      const int aj = mj % na, cj = mj / na;
      const int yj = cj / W, xj = cj - yj * W;
      const float* rj = base + (long long)aj * lv.sa[l] + (long long)yj * lv.sy[l] + (long long)xj * lv.sx[l];
      const float obj = __shfl(my_obj, j, 64);
      float best = -1.0f;
      int bc = 0;
      for (int c = 5 + lane; c < no; c += 64) {
        const float v = hd_sigmoid(rj[c * so]);
        if (v > best) {
          best = v;
          bc = c - 5;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oc = __shfl_xor(bc, o, 64);
        if (ob > best || (ob == best && oc < bc)) {
          best = ob;
          bc = oc;
        }
      }
      if (lane == j) {
        my_conf = obj * best;
        my_cls = bc;
      }
    }
    const bool take = my_conf > conf_thr;
    const unsigned long long mm = __ballot(take);
    if (mm != 0ull) {
      int pos0 = 0;
      if (lane == 0) pos0 = atomicAdd(&w.n_keys[b], __popcll(mm));
      pos0 = __shfl(pos0, 0, 64);
      if (take) {
        // the decode kernel's expressions, in its order
        const float s0 = hd_sigmoid(r[0]), s1 = hd_sigmoid(r[so]), s2 = hd_sigmoid(r[2 * so]), s3 = hd_sigmoid(r[3 * so]);
        const float cx = (s0 * 2.0f - 0.5f + (float)x) * stride, cy = (s1 * 2.0f - 0.5f + (float)y) * stride;
        const float bw = (s2 * 2.0f) * (s2 * 2.0f) * lv.anchor_w[l][a], bh = (s3 * 2.0f) * (s3 * 2.0f) * lv.anchor_h[l][a];
        const int i = ix.first_row[l] + (a * H + y) * W + x;  // row of the decoded layout
        cand_box[(size_t)b * n + i] = make_float4(cx - bw * 0.5f, cy - bh * 0.5f, cx + bw * 0.5f, cy + bh * 0.5f);
        cand_cls[(size_t)b * n + i] = my_cls;
        cand_conf[(size_t)b * n + i] = my_conf;
        keys[pos0 + __popcll(mm & ((1ull << lane) - 1ull))] = make_key(my_conf, (uint32_t)i);
      }
    }
  }
}

struct YoloWs {
  float4* cand_box;
  int32_t* cand_cls;
  float* cand_conf;
  int32_t* keep;
  size_t core_off, bytes;
};

inline YoloWs yolo_layout(void* base, int B, int n, int max_det) {
  YoloWs y;
  size_t off = 0;
  char* p = reinterpret_cast<char*>(base);
  auto take = [&](size_t bytes) {
    char* r = p ? p + off : nullptr;
    off += hd_align_up(bytes, 256);
    return r;
  };
  y.cand_box = reinterpret_cast<float4*>(take((size_t)B * n * sizeof(float4)));
  y.cand_cls = reinterpret_cast<int32_t*>(take((size_t)B * n * sizeof(int32_t)));
  y.cand_conf = reinterpret_cast<float*>(take((size_t)B * n * sizeof(float)));
  y.keep = reinterpret_cast<int32_t*>(take((size_t)B * (max_det > 0 ? max_det : 1) * sizeof(int32_t)));
  y.core_off = off;
  y.bytes = off;
  return y;
}

}  // namespace

extern "C" size_t hd_nms_workspace_bytes(int B, int N, int max_candidates) {
  if (B <= 0 || N <= 0) return 0;
  return core_layout(nullptr, B, N, max_candidates).bytes;
}

extern "C" hd_status hd_nms_batched(const float* boxes, const float* scores, const int32_t* cls,
                                    const int32_t* n_valid, int B, int N, float score_thr, float iou_thr,
                                    int max_candidates, int max_out, int32_t* keep, int32_t* keep_count,
                                    void* workspace, size_t workspace_bytes, hd_stream stream) {
  if (B < 0 || N < 0 || max_out < 0) return HD_ERR_SHAPE;
  if (B == 0) return HD_OK;
  if (B > 65535) return HD_ERR_SHAPE;
  if (!keep_count || (max_out > 0 && !keep)) return HD_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  if (N == 0 || max_out == 0) {
    if (hd_fill32(keep_count, 0u, (size_t)B, s) != HD_OK) return HD_ERR_LAUNCH;
    if (max_out > 0 && hd_fill32(keep, 0xFFFFFFFFu, (size_t)B * max_out, s) != HD_OK)
      return HD_ERR_LAUNCH;
    return HD_OK;
  }
  if (!boxes || !scores) return HD_ERR_NULL;
  if (!hd_aligned16(boxes)) return HD_ERR_ALIGN;
  CoreWs w = core_layout(workspace, B, N, max_candidates);
  if (w.cap > kMaxCand) return HD_ERR_SHAPE;
  if (!workspace || workspace_bytes < w.bytes) return HD_ERR_WORKSPACE;
  if (!hd_aligned16(workspace)) return HD_ERR_ALIGN;
  if (hd_fill32(w.n_keys, 0u, (size_t)B, s) != HD_OK) return HD_ERR_LAUNCH;
  int gx = (N + 255) / 256;
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(nms_keys_kernel, dim3(gx, B), dim3(256), 0, s, scores, n_valid, N, score_thr, w);
  HD_RETURN_IF_LAUNCH_FAILED();
  return run_core(w, B, reinterpret_cast<const float4*>(boxes), cls, (long long)N, iou_thr, max_out, keep,
                  keep_count, s);
}

extern "C" size_t hd_yolo_nms_workspace_bytes(int B, int n, int max_nms, int max_det) {
  if (B <= 0 || n <= 0 || max_det <= 0) return 0;
  YoloWs y = yolo_layout(nullptr, B, n, max_det);
  return y.bytes + core_layout(nullptr, B, n, max_nms).bytes;
}

extern "C" hd_status hd_yolo_nms(const float* dec, int B, int n, int no, float conf_thr, float iou_thr,
                                 int max_nms, int max_det, float* det, int32_t* det_count, void* workspace,
                                 size_t workspace_bytes, hd_stream stream) {
  if (B <= 0 || n <= 0 || no < 6 || max_det <= 0 || max_det > (1 << 20) || B > 65535) return HD_ERR_SHAPE;
  if (!dec || !det || !det_count) return HD_ERR_NULL;
  if (!workspace) return HD_ERR_WORKSPACE;
  if (!hd_aligned16(workspace)) return HD_ERR_ALIGN;
  YoloWs y = yolo_layout(workspace, B, n, max_det);
  CoreWs w = core_layout(reinterpret_cast<char*>(workspace) + y.core_off, B, n, max_nms);
  if (w.cap > kMaxCand) return HD_ERR_SHAPE;
  if (workspace_bytes < y.bytes + w.bytes) return HD_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  if (hd_fill32(w.n_keys, 0u, (size_t)B, s) != HD_OK) return HD_ERR_LAUNCH;
  int gx = (n + 255) / 256;  // 4 waves per block, 64 rows per wave per iteration
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(yolo_cand_kernel, dim3(gx, B), dim3(256), 0, s, dec, n, no, conf_thr, w, y.cand_box, y.cand_cls,
                     y.cand_conf);
  HD_RETURN_IF_LAUNCH_FAILED();
  hd_status st = run_core(w, B, y.cand_box, y.cand_cls, (long long)n, iou_thr, max_det, y.keep, det_count, s);
  if (st != HD_OK) return st;
  hipLaunchKernelGGL(yolo_det_kernel, dim3((max_det + 255) / 256, B), dim3(256), 0, s, y.keep, det_count, n, max_det,
                     y.cand_box, y.cand_cls, y.cand_conf, det);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

static int yolo_total_rows(const hd_yolo_levels* lv, RawIndex* ix) {
  long long n = 0;
  for (int l = 0; l <= HD_MAX_LEVELS; ++l) {
    if (ix) ix->first_row[l] = (int)n;
    if (l < lv->n_levels) n += (long long)lv->na * lv->h[l] * lv->w[l];
  }
  return n > 0x7fffffffLL ? -1 : (int)n;
}

extern "C" size_t hd_yolo_detect_workspace_bytes(const hd_yolo_levels* lv, int B, int max_nms, int max_det) {
  if (!lv || lv->n_levels <= 0 || lv->n_levels > HD_MAX_LEVELS) return 0;
  const int n = yolo_total_rows(lv, nullptr);
  return n > 0 ? hd_yolo_nms_workspace_bytes(B, n, max_nms, max_det) : 0;
}

extern "C" hd_status hd_yolo_detect(const hd_yolo_levels* lv, int B, float conf_thr, float iou_thr, int max_nms,
                                    int max_det, float* det, int32_t* det_count, void* workspace,
                                    size_t workspace_bytes, hd_stream stream) {
  if (!lv) return HD_ERR_NULL;
  if (B <= 0 || max_det <= 0 || max_det > (1 << 20) || B > 65535 || lv->n_levels <= 0 || lv->n_levels > HD_MAX_LEVELS ||
      lv->na <= 0 || lv->na > HD_MAX_ANCHORS || lv->no < 6)
    return HD_ERR_SHAPE;
  for (int l = 0; l < lv->n_levels; ++l) {
    if (!lv->pred[l]) return HD_ERR_NULL;
    if (lv->h[l] <= 0 || lv->w[l] <= 0) return HD_ERR_SHAPE;
  }
  if (!det || !det_count) return HD_ERR_NULL;
  RawIndex ix;
  const int n = yolo_total_rows(lv, &ix);
  if (n <= 0) return HD_ERR_SHAPE;
  if (!workspace) return HD_ERR_WORKSPACE;
  if (!hd_aligned16(workspace)) return HD_ERR_ALIGN;
  YoloWs y = yolo_layout(workspace, B, n, max_det);
  CoreWs w = core_layout(reinterpret_cast<char*>(workspace) + y.core_off, B, n, max_nms);
  if (w.cap > kMaxCand) return HD_ERR_SHAPE;
  if (workspace_bytes < y.bytes + w.bytes) return HD_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  if (hd_fill32(w.n_keys, 0u, (size_t)B, s) != HD_OK) return HD_ERR_LAUNCH;
  int max_rows = 0;
  for (int l = 0; l < lv->n_levels; ++l) max_rows = std::max(max_rows, lv->na * lv->h[l] * lv->w[l]);
  int gx = (max_rows + 255) / 256;
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(yolo_raw_cand_kernel, dim3(gx, B, lv->n_levels), dim3(256), 0, s, *lv, ix, n, conf_thr, w, y.cand_box,
                     y.cand_cls, y.cand_conf);
  HD_RETURN_IF_LAUNCH_FAILED();
  hd_status st = run_core(w, B, y.cand_box, y.cand_cls, (long long)n, iou_thr, max_det, y.keep, det_count, s);
  if (st != HD_OK) return st;
  hipLaunchKernelGGL(yolo_det_kernel, dim3((max_det + 255) / 256, B), dim3(256), 0, s, y.keep, det_count, n, max_det,
                     y.cand_box, y.cand_cls, y.cand_conf, det);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}
