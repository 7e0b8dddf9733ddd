// wbf.hip — Weighted Boxes Fusion for TTA (reference README.md:19; SURVEY.md §8a "Weighted Boxes Fusion").
//
// WBF is sequential by definition (each box joins the best cluster built so far) — but only within a
// label, because a box never joins a cluster of another label.  Parallelism: across images (one
// workgroup each), across labels (one wave per label segment of the (label, confidence)-sorted
// input, no workgroup barrier in the loop), and across the segment's clusters (64 lanes test them
// against the current box, the wave arg-maxes a packed (iou bits, ~cluster) key, lane 0 updates
// the winner).  Latency-bound
//  — so nothing on that chain may wait for HBM: the
// sorted inputs are gathered once into contiguous arrays and the next box is prefetched into
// registers while the current one is clustered, and the cluster state (fused box, running sums,
// count, label: 44 B per cluster) lives in LDS whenever it fits (K <= ~2900; global fallback above).
// 8 images x 1500 boxes of 10 labels (3 TTA views x ~500 objects): 0.26 ms; 2.9 ms when one workgroup
// walked all boxes with two barriers per box.  Sorting reuses the workgroup bitonic sort.
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
  float4* so_box;    // [B, K] inputs in visiting order (gathered once after the sort)
  float* so_conf;    // [B, K]
  int32_t* so_label; // [B, K]
  int32_t* seg;      // [B, K] start position of every label segment (any order)
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
  w.so_box = reinterpret_cast<float4*>(take((size_t)B * K * sizeof(float4)));
  w.so_conf = reinterpret_cast<float*>(take((size_t)B * K * sizeof(float)));
  w.so_label = reinterpret_cast<int32_t*>(take((size_t)B * K * sizeof(int32_t)));
  w.seg = reinterpret_cast<int32_t*>(take((size_t)B * K * sizeof(int32_t)));
  w.bytes = off;
  return w;
}

template <bool LDS_STATE>
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
  extern __shared__ __align__(16) unsigned char s_state[];  // LDS_STATE: [K] box, sum (float4), conf, n, label
  float4* cl_box = LDS_STATE ? reinterpret_cast<float4*>(s_state) : w.cl_box + (size_t)b * K;
  float4* cl_sum = LDS_STATE ? cl_box + K : w.cl_sum + (size_t)b * K;
  float* cl_conf = LDS_STATE ? reinterpret_cast<float*>(cl_sum + K) : w.cl_conf + (size_t)b * K;
  int32_t* cl_n = LDS_STATE ? reinterpret_cast<int32_t*>(cl_conf + K) : w.cl_n + (size_t)b * K;
  int32_t* cl_label = LDS_STATE ? cl_n + K : w.cl_label + (size_t)b * K;
  float4* so_box = w.so_box + (size_t)b * K;
  float* so_conf = w.so_conf + (size_t)b * K;
  int32_t* so_label = w.so_label + (size_t)b * K;
  // 1. candidate keys: (label desc, confidence desc, lower index first).  Clusters only ever match boxes of
  //    their own label, so the visiting order matters within a label only; grouping by label lets every label's
  //    sequence run on its own wave, concurrently, with no workgroup barrier inside the clustering loop.
  if (t == 0) s_n = 0;
  __syncthreads();
  for (int i = t; i < kv; i += blockDim.x) {
    const float sc = scores[(size_t)b * K + i];
    if (!(sc < skip_thr)) {
      const float conf = sc * model_weight[(size_t)b * K + i];
      const int pos = atomicAdd(&s_n, 1);
      keys[pos] = ((u64)(labels[(size_t)b * K + i] & 0xFFF) << 52) | ((u64)hd_f2key(conf) << 20) |
                  (u64)(0xFFFFFu - (uint32_t)i);
    }
  }
  __syncthreads();
  const int n = s_n;
  int n_pow2 = 1;
  while (n_pow2 < n) n_pow2 <<= 1;
  for (int i = n + t; i < n_pow2; i += blockDim.x) keys[i] = 0ull;
  __syncthreads();
  block_sort_desc(keys, n_pow2, s_tile, kWbfTile);
  // 1b. inputs in visiting order, contiguous: the clustering loop never chases an index through HBM
  for (int r = t; r < n; r += blockDim.x) {
    const int i = (int)(0xFFFFFu - (uint32_t)(keys[r] & 0xFFFFFull));
    so_box[r] = bx[i];
    so_conf[r] = scores[(size_t)b * K + i] * model_weight[(size_t)b * K + i];
    so_label[r] = labels[(size_t)b * K + i];
    cl_n[r] = 0;  // slot r is a cluster iff cl_n[r] > 0 afterwards
  }
  if (t == 0) s_n = 0;
  __syncthreads();
  // 1c. label segments (start positions, any order)
  int32_t* seg = w.seg + (size_t)b * K;
  for (int r = t; r < n; r += blockDim.x)
    if (r == 0 || so_label[r] != so_label[r - 1]) seg[atomicAdd(&s_n, 1)] = r;
  __syncthreads();
  const int nseg = s_n;
  // 2. clustering: one wave per label segment, clusters of the segment live in slots [start, start + ncl)
  const int lane = t & 63, wave = t >> 6, nwaves = (int)(blockDim.x >> 6);
  for (int sidx = wave; sidx < nseg; sidx += nwaves) {
    const int start = seg[sidx];
    const int lab = so_label[start];
    int ncl = 0;
    float4 nbox = so_box[start];
    float nconf = so_conf[start];
    int nlab = lab;
    for (int r = start; r < n && nlab == lab; ++r) {
      const float4 box = nbox;
      const float conf = nconf;
      if (r + 1 < n) {  // request box r+1 before clustering box r
        nbox = so_box[r + 1];
        nconf = so_conf[r + 1];
        nlab = so_label[r + 1];
      } else {
        nlab = -1;
      }
      const float barea = hd_box_area(box);
      u64 best = 0ull;
      for (int c = lane; c < ncl; c += 64) {
        const float4 cb = cl_box[start + c];
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
      if (lane == 0) {
        int c;
        float4 sum;
        float sconf;
        int cn;
        if (best == 0ull) {
          c = ncl;
          sum = make_float4(0.f, 0.f, 0.f, 0.f);
          sconf = 0.0f;
          cn = 0;
          cl_label[start + c] = lab;
        } else {
          c = (int)(0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFull));
          sum = cl_sum[start + c];
          sconf = cl_conf[start + c];
          cn = cl_n[start + c];
        }
        sum.x += conf * box.x;
        sum.y += conf * box.y;
        sum.z += conf * box.z;
        sum.w += conf * box.w;
        sconf += conf;
        cl_sum[start + c] = sum;
        cl_conf[start + c] = sconf;
        cl_n[start + c] = cn + 1;
        cl_box[start + c] = make_float4(sum.x / sconf, sum.y / sconf, sum.z / sconf, sum.w / sconf);
      }
      if (best == 0ull) ++ncl;  // wave-uniform
      __threadfence_block();    // lane 0's cluster update is visible to the wave's next scan
    }
  }
  __syncthreads();
  // 3. fused scores, final order: score desc, label asc, older cluster first (= lower slot within a label)
  if (t == 0) s_n = 0;
  __syncthreads();
  int mine = 0;
  for (int c = t; c < n_pow2; c += blockDim.x) {
    u64 key = 0ull;
    if (c < n && cl_n[c] > 0) {
      const float nn = (float)cl_n[c];
      const float sc = (cl_conf[c] / nn) * fminf(weight_sum, nn) / weight_sum;
      const uint32_t tie = ((uint32_t)cl_label[c] << 20) | (uint32_t)c;
      key = ((u64)hd_f2key(sc) << 32) | (u64)(0xFFFFFFFFu - tie);
      cl_conf[c] = sc;  // running sum no longer needed
      ++mine;
    }
    keys[c] = key;
  }
  if (mine) atomicAdd(&s_n, mine);
  __syncthreads();
  const int ncl_all = s_n;
  block_sort_desc(keys, n_pow2, s_tile, kWbfTile);
  for (int k = t; k < K; k += blockDim.x) {
    float* o = out + ((size_t)b * K + k) * 6;
    if (k < ncl_all) {
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
  if (t == 0) out_count[b] = ncl_all;
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
  // cluster state in LDS when it fits beside the 32 KiB sort tile (160 KiB per CU): 44 B per possible cluster
  const size_t state_bytes = (size_t)K * (2 * sizeof(float4) + sizeof(float) + 2 * sizeof(int32_t));
  if (state_bytes <= (size_t)120 * 1024) {
    if (state_bytes > (size_t)32 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&wbf_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)state_bytes) != hipSuccess)
      return HD_ERR_LAUNCH;
    hipLaunchKernelGGL(wbf_kernel<true>, dim3(B), dim3(1024), state_bytes, s, reinterpret_cast<const float4*>(boxes),
                       scores, labels, model_weight, k_valid, K, weight_sum, iou_thr, skip_thr, w, out, out_count);
  } else {
    hipLaunchKernelGGL(wbf_kernel<false>, dim3(B), dim3(1024), 0, s, reinterpret_cast<const float4*>(boxes), scores,
                       labels, model_weight, k_valid, K, weight_sum, iou_thr, skip_thr, w, out, out_count);
  }
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}
