// wbf.hip — Weighted Boxes Fusion for TTA (reference README.md:19; SURVEY.md §8a "Weighted Boxes Fusion").
//
// WBF is sequential by definition (each box joins the best cluster built so far), so the
// parallelism is across images (one workgroup each) and, inside an image, across the existing
// clusters: every lane tests a slice of the clusters against the current box, the workgroup
// arg-maxes a packed (iou bits, ~cluster) key, one lane updates the winner.  Latency-bound
// (SURVEY.md §8d), a few barriers per input box.  Sorting reuses the workgroup bitonic sort.
//
// Arithmetic order (running sums, divisions) is the oracle's orc_wbf(), so fused boxes and
// scores are bit-identical and the output order (score desc, label asc, older cluster first)
// is reproduced exactly.  Limits: labels < 4096, K <= 2^20.
#include "hd_nms_core.h"

using namespace hdnms;

namespace {

struct WbfWs {
  u64* keys;         // [B, Kpad]
  float4* cl_box;    // [B, K] fused boxes
  float4* cl_sum;    // [B, K] running sums of conf * coords
  float* cl_conf;    // [B, K]
  int32_t* cl_n;     // [B, K]
  int32_t* cl_label; // [B, K]
  int Kpad;
  size_t bytes;
};

inline WbfWs wbf_layout(void* base, int B, int K) {
  WbfWs w;
  w.Kpad = hd_next_pow2(K < 1 ? 1 : K);
  size_t off = 0;
  char* p = reinterpret_cast<char*>(base);
  auto take = [&](size_t bytes) {
    char* r = p ? p + off : nullptr;
    off += hd_align_up(bytes, 256);
    return r;
  };
  w.keys = reinterpret_cast<u64*>(take((size_t)B * w.Kpad * sizeof(u64)));
  w.cl_box = reinterpret_cast<float4*>(take((size_t)B * K * sizeof(float4)));
  w.cl_sum = reinterpret_cast<float4*>(take((size_t)B * K * sizeof(float4)));
  w.cl_conf = reinterpret_cast<float*>(take((size_t)B * K * sizeof(float)));
  w.cl_n = reinterpret_cast<int32_t*>(take((size_t)B * K * sizeof(int32_t)));
  w.cl_label = reinterpret_cast<int32_t*>(take((size_t)B * K * sizeof(int32_t)));
  w.bytes = off;
  return w;
}

__global__ __launch_bounds__(1024) void wbf_kernel(const float4* __restrict__ boxes, const float* __restrict__ scores,
                                                   const int32_t* __restrict__ labels,
                                                   const float* __restrict__ model_weight,
                                                   const int32_t* __restrict__ k_valid, int K, float weight_sum,
                                                   float iou_thr, float skip_thr, WbfWs w, float* __restrict__ out,
                                                   int32_t* __restrict__ out_count) {
  constexpr int kWbfTile = 4096;  // static 32 KiB tile (the NMS engine uses a larger dynamic one)
  __shared__ u64 s_tile[kWbfTile];
  __shared__ u64 s_red[16];
  __shared__ int s_n;
  const int b = blockIdx.x, t = threadIdx.x;
  const int kv = k_valid ? min(k_valid[b], K) : K;
  u64* keys = w.keys + (size_t)b * w.Kpad;
  const float4* bx = boxes + (size_t)b * K;
  float4* cl_box = w.cl_box + (size_t)b * K;
  float4* cl_sum = w.cl_sum + (size_t)b * K;
  float* cl_conf = w.cl_conf + (size_t)b * K;
  int32_t* cl_n = w.cl_n + (size_t)b * K;
  int32_t* cl_label = w.cl_label + (size_t)b * K;
  // 1. candidate keys (order inside the buffer is irrelevant: keys are unique and get sorted)
  if (t == 0) s_n = 0;
  __syncthreads();
  for (int i = t; i < kv; i += blockDim.x) {
    const float sc = scores[(size_t)b * K + i];
    if (!(sc < skip_thr)) {
      const float conf = sc * model_weight[(size_t)b * K + i];
      const int pos = atomicAdd(&s_n, 1);
      keys[pos] = make_key(conf, (uint32_t)i);
    }
  }
  __syncthreads();
  const int n = s_n;
  int n_pow2 = 1;
  while (n_pow2 < n) n_pow2 <<= 1;
  for (int i = n + t; i < n_pow2; i += blockDim.x) keys[i] = 0ull;
  __syncthreads();
  block_sort_desc(keys, n_pow2, s_tile, kWbfTile);
  // 2. sequential clustering
  int ncl = 0;
  for (int r = 0; r < n; ++r) {
    const int i = (int)key_index(keys[r]);
    const float4 box = bx[i];
    const float conf = scores[(size_t)b * K + i] * model_weight[(size_t)b * K + i];
    const int lab = labels[(size_t)b * K + i];
    const float barea = hd_box_area(box);
    u64 best = 0ull;
    for (int c = t; c < ncl; c += blockDim.x) {
      if (cl_label[c] != lab) continue;
      const float4 cb = cl_box[c];
      const float v = hd_iou_full(cb, hd_box_area(cb), box, barea);
      if (v > iou_thr) {
        const u64 key = ((u64)__float_as_uint(v) << 32) | (u64)(0xFFFFFFFFu - (uint32_t)c);
        if (key > best) best = key;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const u64 other = __shfl_xor(best, o, 64);
      if (other > best) best = other;
    }
    if ((t & 63) == 0) s_red[t >> 6] = best;
    __syncthreads();
    if (t == 0) {
      u64 m = 0ull;
      for (int k = 0; k < (int)(blockDim.x >> 6); ++k)
        if (s_red[k] > m) m = s_red[k];
      int c;
      float4 sum;
      float sconf;
      int cn;
      if (m == 0ull) {
        c = ncl;
        sum = make_float4(0.f, 0.f, 0.f, 0.f);
        sconf = 0.0f;
        cn = 0;
        cl_label[c] = lab;
      } else {
        c = (int)(0xFFFFFFFFu - (uint32_t)(m & 0xFFFFFFFFull));
        sum = cl_sum[c];
        sconf = cl_conf[c];
        cn = cl_n[c];
      }
      sum.x += conf * box.x;
      sum.y += conf * box.y;
      sum.z += conf * box.z;
      sum.w += conf * box.w;
      sconf += conf;
      cl_sum[c] = sum;
      cl_conf[c] = sconf;
      cl_n[c] = cn + 1;
      cl_box[c] = make_float4(sum.x / sconf, sum.y / sconf, sum.z / sconf, sum.w / sconf);
      s_n = (m == 0ull) ? ncl + 1 : ncl;
    }
    __syncthreads();
    ncl = s_n;
  }
  // 3. fused scores, final order: score desc, label asc, older cluster first
  int c_pow2 = 1;
  while (c_pow2 < ncl) c_pow2 <<= 1;
  for (int c = t; c < c_pow2; c += blockDim.x) {
    u64 key = 0ull;
    if (c < ncl) {
      const float nn = (float)cl_n[c];
      const float sc = (cl_conf[c] / nn) * fminf(weight_sum, nn) / weight_sum;
      const uint32_t tie = ((uint32_t)cl_label[c] << 20) | (uint32_t)c;
      key = ((u64)hd_f2key(sc) << 32) | (u64)(0xFFFFFFFFu - tie);
      cl_conf[c] = sc;  // running sum no longer needed
    }
    keys[c] = key;
  }
  __syncthreads();
  block_sort_desc(keys, c_pow2, s_tile, kWbfTile);
  for (int k = t; k < K; k += blockDim.x) {
    float* o = out + ((size_t)b * K + k) * 6;
    if (k < ncl) {
      const uint32_t tie = 0xFFFFFFFFu - (uint32_t)(keys[k] & 0xFFFFFFFFull);
      const int c = (int)(tie & 0xFFFFFu);
      const float4 fb = cl_box[c];
      o[0] = fb.x; o[1] = fb.y; o[2] = fb.z; o[3] = fb.w;
      o[4] = cl_conf[c];
      o[5] = (float)cl_label[c];
    } else {
      o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = 0.0f;
    }
  }
  if (t == 0) out_count[b] = ncl;
}

}  // namespace

extern "C" size_t hd_wbf_workspace_bytes(int B, int K) {
  if (B <= 0 || K <= 0) return 0;
  return wbf_layout(nullptr, B, K).bytes;
}

extern "C" hd_status hd_wbf(const float* boxes, const float* scores, const int32_t* labels,
                            const float* model_weight, const int32_t* k_valid, int B, int K, float weight_sum,
                            float iou_thr, float skip_thr, float* out, int32_t* out_count, void* workspace,
                            size_t workspace_bytes, hd_stream stream) {
  if (B < 0 || K < 0 || K > (1 << 20) || !(weight_sum > 0.0f)) return HD_ERR_SHAPE;
  if (B == 0) return HD_OK;
  if (!out_count) return HD_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  if (K == 0) {
    if (hd_fill32(out_count, 0u, (size_t)B, s) != HD_OK) return HD_ERR_LAUNCH;
    return HD_OK;
  }
  if (!boxes || !scores || !labels || !model_weight || !out) return HD_ERR_NULL;
  if (!hd_aligned16(boxes)) return HD_ERR_ALIGN;
  WbfWs w = wbf_layout(workspace, B, K);
  if (!workspace || workspace_bytes < w.bytes) return HD_ERR_WORKSPACE;
  if (!hd_aligned16(workspace)) return HD_ERR_ALIGN;
  hipLaunchKernelGGL(wbf_kernel, dim3(B), dim3(1024), 0, s, reinterpret_cast<const float4*>(boxes), scores, labels,
                     model_weight, k_valid, K, weight_sum, iou_thr, skip_thr, w, out, out_count);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}
