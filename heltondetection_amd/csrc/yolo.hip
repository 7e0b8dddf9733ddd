// yolo.hip — YOLOv5 head ops for gfx950 (reference README.md:9; SURVEY.md §8a rows
// "YOLOv5 grid decode", "YOLOv5 target assignment", "CIoU loss fwd + bwd").
//
// Layout.  Each level's raw head output is viewed as [B, na, H, W, no] through explicit element
// strides (hd_yolo_levels), so the channels-last conv output [B, H, W, na*no] is consumed in
// place — no permute/contiguous copy.  The fast paths assume so == 1 (the `no` channels of one
// cell are contiguous, true for channels-last); other layouts run the same code, slower.
//
// Kernels (all HBM- or latency-bound, no MFMA):
//   yolo_decode_kernel      read 4*no B + write 4*no B per cell; each lane produces 4 consecutive
//                           floats of the dense [B, n, no] output and stores them as one 16-byte
//                           word (1 KiB contiguous per wave store).
//   yolo_assign_kernel      one lane per slot (level, k, a, b, t); pure index arithmetic.
//   yolo_pos_fwd/bwd        a wave walks 64 slots, finds the matched ones with one ballot and
//                           then spends all 64 lanes on the `no` channels of each matched cell
//                           (coalesced 4*no-byte gathers); wave reductions, one atomic per wave.
//   yolo_obj_fwd            one lane per cell, objectness BCE against tobj, wave+LDS reduction.
//   yolo_grad_dense_kernel  writes the dense gradient (zeros + objectness term) with 16-byte
//                           stores in memory order — the 4*B*n*no-byte write that bounds backward.
#include "hd_ciou.h"
#include "hd_common.h"

namespace {

struct LevelIndex {
  long long first_row[HD_MAX_LEVELS + 1];   // rows (= na*H*W) before each level, per image
  long long first_cell[HD_MAX_LEVELS + 1];  // cells (= B*na*H*W) before each level
};

inline LevelIndex make_index(const hd_yolo_levels& lv, int B) {
  LevelIndex ix;
  long long r = 0, c = 0;
  for (int l = 0; l <= HD_MAX_LEVELS; ++l) {
    ix.first_row[l] = r;
    ix.first_cell[l] = c;
    if (l < lv.n_levels) {
      r += (long long)lv.na * lv.h[l] * lv.w[l];
      c += (long long)B * lv.na * lv.h[l] * lv.w[l];
    }
  }
  return ix;
}

inline hd_status check_levels(const hd_yolo_levels* lv, int B) {
  if (!lv) return HD_ERR_NULL;
  if (lv->n_levels <= 0 || lv->n_levels > HD_MAX_LEVELS || lv->na <= 0 || lv->na > HD_MAX_ANCHORS || lv->no < 5 ||
      B <= 0)
    return HD_ERR_SHAPE;
  for (int l = 0; l < lv->n_levels; ++l) {
    if (!lv->pred[l]) return HD_ERR_NULL;
    if (lv->h[l] <= 0 || lv->w[l] <= 0 || !(lv->stride[l] > 0.0f)) return HD_ERR_SHAPE;
    if ((long long)B * lv->na * lv->h[l] * lv->w[l] > 0x7fffffffLL) return HD_ERR_SHAPE;
  }
  return HD_OK;
}

inline bool is_dense_nhwc_decl(const hd_yolo_levels& g, int l, int B) {
  const long long no = g.no, na = g.na, W = g.w[l], H = g.h[l];
  return g.so[l] == 1 && g.sa[l] == no && g.sx[l] == na * no && g.sy[l] == W * na * no &&
         (B == 1 || g.sb[l] == H * W * na * no);
}

// ------------------------------------------------------------------------------------ decode
//
// One workgroup = 64 consecutive output rows of one (level, image, anchor) group, staged through
// an LDS tile: the gather side reads each input row (`no` contiguous floats when so == 1) with
// consecutive lanes on consecutive channels, applies the decode formula, and drops the value at
// its output position in LDS; the store side then streams the tile out linearly with 16-byte
// stores (1 KiB per wave store).  Per-row index work (y, x, input offset) is done once per row
// by 64 lanes, not per element; the element -> (row, channel) split is one multiply.
constexpr int kDecRows = 64;

struct DecodePlan {
  int first_block[HD_MAX_LEVELS + 1];  // first workgroup of each level
  int tiles[HD_MAX_LEVELS];            // ceil(H*W / kDecRows)
};

__global__ __launch_bounds__(256) void yolo_decode_kernel(const hd_yolo_levels lv, const LevelIndex ix,
                                                          const DecodePlan plan, int B, float* __restrict__ out) {
  extern __shared__ float s_tile[];  // [kDecRows * no] + row tables
  const int no = lv.no, na = lv.na;
  long long* s_off = reinterpret_cast<long long*>(s_tile + (size_t)kDecRows * no + (((size_t)kDecRows * no) & 1));
  int* s_x = reinterpret_cast<int*>(s_off + kDecRows);
  int* s_y = s_x + kDecRows;
  int l = 0;
#pragma unroll
  for (int k = 1; k < HD_MAX_LEVELS; ++k)
    if (k < lv.n_levels && (int)blockIdx.x >= plan.first_block[k]) l = k;
  int rem = (int)blockIdx.x - plan.first_block[l];
  const int tile = rem % plan.tiles[l];
  rem /= plan.tiles[l];
  const int a = rem % na;
  const int b = rem / na;
  const int H = lv.h[l], W = lv.w[l], HW = H * W;
  const int row0 = tile * kDecRows;
  const int rows = min(kDecRows, HW - row0);
  if (threadIdx.x < rows) {
    const int rr = row0 + threadIdx.x;
    const int y = rr / W, x = rr - y * W;
    s_x[threadIdx.x] = x;
    s_y[threadIdx.x] = y;
    s_off[threadIdx.x] = b * lv.sb[l] + a * lv.sa[l] + y * lv.sy[l] + x * lv.sx[l];
  }
  __syncthreads();
  const float* __restrict__ pr = lv.pred[l];
  const long long so = lv.so[l];
  const float stride = lv.stride[l], aw = lv.anchor_w[l][a], ah = lv.anchor_h[l][a];
  const float inv_no = 1.0f / (float)no;
  const int elems = rows * no;
  // 4 elements per lane per iteration: the four (row table lookup -> global load) chains are issued
  // before any value is used, so four 4-byte loads per lane are in flight instead of one
  for (int i0 = threadIdx.x; i0 < elems; i0 += blockDim.x * 4) {
    int r[4], c[4];
    float raw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * blockDim.x;
      int rr = (int)(((float)i + 0.5f) * inv_no);
      int cc = i - rr * no;
      if (cc < 0) { --rr; cc += no; }
      if (cc >= no) { ++rr; cc -= no; }
      r[u] = rr;
      c[u] = cc;
      raw[u] = i < elems ? pr[s_off[rr] + cc * so] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * blockDim.x;
      if (i >= elems) break;
      const float s = hd_sigmoid(raw[u]);
      float v;
      if (c[u] == 0) v = (s * 2.0f - 0.5f + (float)s_x[r[u]]) * stride;
      else if (c[u] == 1) v = (s * 2.0f - 0.5f + (float)s_y[r[u]]) * stride;
      else if (c[u] == 2) v = (s * 2.0f) * (s * 2.0f) * aw;
      else if (c[u] == 3) v = (s * 2.0f) * (s * 2.0f) * ah;
      else v = s;
      s_tile[i] = v;
    }
  }
  __syncthreads();
  const long long n_total = ix.first_row[lv.n_levels];
  float* __restrict__ o = out + (((long long)b * n_total + ix.first_row[l] + (long long)a * HW + row0) * no);
  if ((reinterpret_cast<uintptr_t>(o) & 15u) == 0) {
    const int words = elems >> 2;
    const float4* t4 = reinterpret_cast<const float4*>(s_tile);
    float4* o4 = reinterpret_cast<float4*>(o);
    for (int q = threadIdx.x; q < words; q += blockDim.x) o4[q] = t4[q];
    for (int i = (words << 2) + threadIdx.x; i < elems; i += blockDim.x) o[i] = s_tile[i];
  } else {
    for (int i = threadIdx.x; i < elems; i += blockDim.x) o[i] = s_tile[i];
  }
}

// Dense channels-last fast path.  The level's tensor is one contiguous block in memory order
// (b, y, x, a, c); a workgroup takes kDenseCells consecutive cells = kDenseCells*na*no contiguous floats,
// reads them with 16-byte loads (every byte of every cache line is used exactly once, unlike the
// per-anchor gather above, whose 4*no-byte rows sit at a 4*na*no-byte stride), decodes each value
// in registers and drops it at its OUTPUT position in an LDS tile laid out [a][cell][c]; each
// anchor's kDenseCells*no floats are then contiguous in the output and are streamed out with
// 16-byte stores when aligned.
constexpr int kDenseCells = 16;

__global__ __launch_bounds__(256) void yolo_decode_dense_kernel(const hd_yolo_levels lv, const LevelIndex ix,
                                                                const DecodePlan plan, int B,
                                                                float* __restrict__ out) {
  extern __shared__ float s_tile[];  // [na][kDenseCells][no]
  __shared__ float s_gx[kDenseCells], s_gy[kDenseCells];
  __shared__ float s_aw[HD_MAX_ANCHORS], s_ah[HD_MAX_ANCHORS];  // per-lane anchor index: keep the table out of the kernarg struct
  const int no = lv.no, na = lv.na;
  int l = 0;
#pragma unroll
  for (int k = 1; k < HD_MAX_LEVELS; ++k)
    if (k < lv.n_levels && (int)blockIdx.x >= plan.first_block[k]) l = k;
  int rem = (int)blockIdx.x - plan.first_block[l];
  const int tile = rem % plan.tiles[l];
  const int b = rem / plan.tiles[l];
  const int H = lv.h[l], W = lv.w[l], HW = H * W;
  const int cell0 = tile * kDenseCells;
  const int cells = min(kDenseCells, HW - cell0);
  const int row_len = na * no;               // floats per cell
  const int elems = cells * row_len;
  const float* __restrict__ src = lv.pred[l] + (size_t)b * lv.sb[l] + (size_t)cell0 * row_len;
  const float stride = lv.stride[l];
  if (threadIdx.x < cells) {
    const int gcell = cell0 + threadIdx.x;
    const int y = gcell / W;
    s_gx[threadIdx.x] = (float)(gcell - y * W);
    s_gy[threadIdx.x] = (float)y;
  }
  if (threadIdx.x >= 64 && threadIdx.x < 64 + na) {
    s_aw[threadIdx.x - 64] = lv.anchor_w[l][threadIdx.x - 64];
    s_ah[threadIdx.x - 64] = lv.anchor_h[l][threadIdx.x - 64];
  }
  __syncthreads();
  // lane-consecutive elements: 256-byte coalesced loads and conflict-free LDS writes (consecutive lanes
  // hit consecutive words).  The (cell, anchor, channel) digits of a lane's element are computed once and
  // then advanced by the block stride with carries — no division in the loop.
  int c = threadIdx.x % no;
  int t = threadIdx.x / no;
  int a = t % na;
  int cell = t / na;
  const int dc = (int)blockDim.x % no;
  const int t2 = (int)blockDim.x / no;
  const int da = t2 % na, dcell = t2 / na;
  for (int i0 = threadIdx.x; i0 < elems; i0 += blockDim.x * 4) {
    float raw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * (int)blockDim.x;
      raw[u] = i < elems ? src[i] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * (int)blockDim.x;
      if (i < elems) {
        const float s = hd_sigmoid(raw[u]);
        float v = s;
        if (c < 4) {
          if (c == 0) v = (s * 2.0f - 0.5f + s_gx[cell]) * stride;
          else if (c == 1) v = (s * 2.0f - 0.5f + s_gy[cell]) * stride;
          else if (c == 2) v = (s * 2.0f) * (s * 2.0f) * s_aw[a];
          else v = (s * 2.0f) * (s * 2.0f) * s_ah[a];
        }
        s_tile[(a * kDenseCells + cell) * no + c] = v;
      }
      // advance the digits by blockDim.x
      c += dc;
      int carry = c >= no;
      c -= carry * no;
      a += da + carry;
      carry = a >= na;
      a -= carry * na;
      carry += a >= na;  // da + carry can wrap twice only if na == 1; handled by the second compare
      a -= (a >= na) * na;
      cell += dcell + carry;
    }
  }
  __syncthreads();
  const long long n_total = ix.first_row[lv.n_levels];
  const int per_a = cells * no;
  for (int a = 0; a < na; ++a) {
    float* __restrict__ o = out + (((long long)b * n_total + ix.first_row[l] + (long long)a * HW + cell0) * no);
    const float* __restrict__ t = s_tile + (size_t)a * kDenseCells * no;
    if ((reinterpret_cast<uintptr_t>(o) & 15u) == 0) {
      const int words = per_a >> 2;
      for (int q = threadIdx.x; q < words; q += blockDim.x)
        reinterpret_cast<float4*>(o)[q] = reinterpret_cast<const float4*>(t)[q];
      for (int i = (words << 2) + threadIdx.x; i < per_a; i += blockDim.x) o[i] = t[i];
    } else {
      for (int i = threadIdx.x; i < per_a; i += blockDim.x) o[i] = t[i];
    }
  }
}

// ------------------------------------------------------------------------------------ assign

__global__ __launch_bounds__(256) void yolo_assign_kernel(const hd_yolo_levels lv, const float4* __restrict__ gt_boxes,
                                                          const int32_t* __restrict__ gt_cls,
                                                          const int32_t* __restrict__ m_valid, int B, int M,
                                                          float anchor_t, int32_t* __restrict__ slot_cell,
                                                          float4* __restrict__ slot_tbox,
                                                          int32_t* __restrict__ slot_cls) {
  const int na = lv.na;
  const long long S = (long long)5 * na * B * M;
  const long long total = S * lv.n_levels;
  for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total;
       o += (long long)gridDim.x * blockDim.x) {
    const int l = (int)(o / S);
    long long s = o - (long long)l * S;
    const int t = (int)(s % M);
    s /= M;
    const int b = (int)(s % B);
    s /= B;
    const int a = (int)(s % na);
    const int k = (int)(s / na);
    int32_t cell = -1, cls = 0;
    float4 tb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < m_valid[b]) {
      const int H = lv.h[l], W = lv.w[l];
      const float st = lv.stride[l];
      const float4 g = gt_boxes[(long long)b * M + t];
      const float gx = ((g.x + g.z) * 0.5f) / st, gy = ((g.y + g.w) * 0.5f) / st;
      const float gw = (g.z - g.x) / st, gh = (g.w - g.y) / st;
      const float aw = lv.anchor_w[l][a] / st, ah = lv.anchor_h[l][a] / st;
      const float rw = gw / aw, rh = gh / ah;
      const float m1 = fmaxf(rw, 1.0f / rw), m2 = fmaxf(rh, 1.0f / rh);
      if (fmaxf(m1, m2) < anchor_t) {
        const float fx = gx - floorf(gx), fy = gy - floorf(gy);
        const float ix = (float)W - gx, iy = (float)H - gy;
        const float fix = ix - floorf(ix), fiy = iy - floorf(iy);
        bool sel;
        float ox = 0.0f, oy = 0.0f;
        switch (k) {
          case 0: sel = true; break;
          case 1: sel = (fx < 0.5f) && (gx > 1.0f); ox = 0.5f; break;
          case 2: sel = (fy < 0.5f) && (gy > 1.0f); oy = 0.5f; break;
          case 3: sel = (fix < 0.5f) && (ix > 1.0f); ox = -0.5f; break;
          default: sel = (fiy < 0.5f) && (iy > 1.0f); oy = -0.5f; break;
        }
        if (sel) {
          int gi = (int)(gx - ox), gj = (int)(gy - oy);
          gi = gi < 0 ? 0 : (gi > W - 1 ? W - 1 : gi);
          gj = gj < 0 ? 0 : (gj > H - 1 ? H - 1 : gj);
          cell = (int32_t)((((long long)b * na + a) * H + gj) * W + gi);
          cls = gt_cls[(long long)b * M + t];
          tb = make_float4(gx - (float)gi, gy - (float)gj, gw, gh);
        }
      }
    }
    slot_cell[o] = cell;
    slot_cls[o] = cls;
    slot_tbox[o] = tb;
  }
}

// ------------------------------------------------------------------------------------ loss
//
// tobj (objectness targets) lives in the workspace, one float per cell.  Its index order per
// level follows the MEMORY order of the prediction tensor when that tensor is dense channels-last
// ([B,H,W,na,no] contiguous: index m = ((b*H + y)*W + x)*na + a, so the objectness logit of cell
// m sits at pred[m*no + 4] and both the objectness pass and the dense gradient pass stream
// linearly with no index decomposition), and the logical order ((b*na + a)*H + y)*W + x
// otherwise.  Forward and backward derive the flag from the same `lv`, so they always agree.

struct LevelFlags {
  int pred_dense[HD_MAX_LEVELS];  // prediction tensor: 0 strided, 1 dense channels-last, 2 dense with PADDED pixel rows
  int grad_dense[HD_MAX_LEVELS];  // the same of the gradient tensor (bwd only)
};

// 1: [B,H,W,na,no] contiguous.  2: the same with every pixel's na*no floats followed by padding up to a pitch of sx floats
// (sx % 4 == 0: the rows of hd_head_conv_fwd, 255 -> 256) — cell m = pixel * na + a then sits at pixel * sx + a * no, the
// dense kernels stream whole padded rows (the gradient's padding columns are written as zeros), and tobj keeps the dense
// cell order.  0: anything else (generic strided code).
inline int is_dense_nhwc(const hd_yolo_levels& g, int l, int B) {
  const long long no = g.no, na = g.na, W = g.w[l], H = g.h[l], P = g.sx[l];
  if (!(g.so[l] == 1 && g.sa[l] == no && g.sy[l] == W * P && (B == 1 || g.sb[l] == H * W * P) && hd_aligned16(g.pred[l])))
    return 0;
  if (P == na * no) return 1;
  return (P > na * no && P % 4 == 0) ? 2 : 0;
}

// Loss accumulators (sum(1-ciou), n_pos, sum bce_cls, sum bce_obj per level).  The totals sit in the
// first 32 floats of the workspace; workgroups add into one of kAccRep replicas a cache line apart
// (replica = blockIdx.x % kAccRep) and yolo_loss_finalize_kernel folds them.  With a single copy all
// n_levels*4 floats share one 128-byte line and every wave of the launch serialises on that L2 line:
// measured 71.7 us for the matched-cell pass with one atomic triple per wave, against 24 us of work.
constexpr int kAccRep = 32;
constexpr int kAccStride = 32;  // floats = 128 bytes >= HD_MAX_LEVELS*4 floats
__device__ __forceinline__ float* acc_replica(float* acc, unsigned block) {
  return acc + 64 + (block % kAccRep) * kAccStride;
}

struct CellPos {
  int b, a, gj, gi;
};

__device__ __forceinline__ CellPos cell_decompose(const hd_yolo_levels& lv, int l, int cell) {
  const int W = lv.w[l], H = lv.h[l];
  CellPos c;
  c.gi = cell % W;
  int r = cell / W;
  c.gj = r % H;
  r /= H;
  c.a = r % lv.na;
  c.b = r / lv.na;
  return c;
}

__device__ __forceinline__ long long cell_offset(const hd_yolo_levels& lv, int l, const CellPos c) {
  return c.b * lv.sb[l] + c.a * lv.sa[l] + c.gj * lv.sy[l] + c.gi * lv.sx[l];
}

__device__ __forceinline__ int tobj_index(const hd_yolo_levels& lv, int l, int dense, const CellPos c, int cell) {
  return dense ? ((c.b * lv.h[l] + c.gj) * lv.w[l] + c.gi) * lv.na + c.a : cell;
}

// box part of one positive: sigmoids, xywh (cell units) -> xyxy
__device__ __forceinline__ float4 yolo_pbox(const float t0, const float t1, const float t2, const float t3,
                                            float aw, float ah, float* sg) {
  sg[0] = hd_sigmoid(t0);
  sg[1] = hd_sigmoid(t1);
  sg[2] = hd_sigmoid(t2);
  sg[3] = hd_sigmoid(t3);
  const float px = sg[0] * 2.0f - 0.5f, py = sg[1] * 2.0f - 0.5f;
  const float pw = (sg[2] * 2.0f) * (sg[2] * 2.0f) * aw, ph = (sg[3] * 2.0f) * (sg[3] * 2.0f) * ah;
  return make_float4(px - pw * 0.5f, py - ph * 0.5f, px + pw * 0.5f, py + ph * 0.5f);
}

__device__ __forceinline__ float4 tbox_xyxy(const float4 tb) {
  return make_float4(tb.x - tb.z * 0.5f, tb.y - tb.w * 0.5f, tb.x + tb.z * 0.5f, tb.y + tb.w * 0.5f);
}

// Gradient element types.  float: the gradient tensor of hd_yolo_levels as declared.  GradBf16 (hd_yolo_loss_bwd_*_bf16): the
// same element strides over bfloat16 elements — the padded rows hd_head_conv_bwd consumes, written without a float32 copy in
// between.  A matched cell's box / class terms are ADDED (a cell can be matched by several targets): float32 atomics there,
// and for bf16 the packed bfloat16 atomic add of gfx950 on the 32-bit word holding the element, every addition rounded (what a
// half-precision index_put_(accumulate=True) does).
struct GradF32 {
  typedef float elem;
  __device__ static __forceinline__ void add(float* base, long long e, float v) { atomicAdd(base + e, v); }
};
struct GradBf16 {
  typedef unsigned short elem;
  __device__ static __forceinline__ unsigned short round_bf16(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
  __device__ static __forceinline__ void add(float* base_as_float, long long e, float v) {
    // global_atomic_pk_add_bf16 on the 32-bit word that holds the element: the value in its half, +0 in the other (x + 0 = x),
    // no return value, no retry loop (a compare-and-swap version made the matched-cell kernel 14 us slower: neighbouring
    // lanes share words and every collision cost a round trip)
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(1))) s16x2 global_s16x2;
    unsigned short* base = reinterpret_cast<unsigned short*>(base_as_float);
    const short b = (short)round_bf16(v);
    const s16x2 val = (e & 1ll) ? s16x2{(short)0, b} : s16x2{b, (short)0};
    __builtin_amdgcn_global_atomic_fadd_v2bf16((global_s16x2*)(base + (e & ~1ll)), val);
  }
};

// Matched-cell terms.  grid: (ceil(S / (4 waves * 4 slots)), n_levels).  A wave owns 4 consecutive
// slots (short serial chain: the per-slot work is a dependent gather + CIoU), finds the matched
// ones with one ballot and spends all 64 lanes on the `no` channels of each matched cell.
// Backward accumulates the bias gradient (sum over cells, per channel) in LDS first and issues
// one global atomic per touched channel per workgroup.
constexpr int kSlotsPerWave = 4;

template <bool BWD, bool FOCAL, typename G = GradF32>
__global__ __launch_bounds__(256) void yolo_pos_kernel(const hd_yolo_levels lv, const hd_yolo_levels glv,
                                                       const LevelIndex ix, const LevelFlags fl, int B, long long S,
                                                       const int32_t* __restrict__ slot_cell,
                                                       const float4* __restrict__ slot_tbox,
                                                       const int32_t* __restrict__ slot_cls,
                                                       const hd_yolo_loss_cfg cfg,
                                                       const float* __restrict__ grad_scale, float* __restrict__ acc,
                                                       float* __restrict__ tobj, float* __restrict__ grad_bias,
                                                       const float* __restrict__ obj_part, int n_part) {
  extern __shared__ float s_gb[];  // [na*no] bias-gradient partials (BWD with grad_bias only)
  const int l = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int no = lv.no, nc = lv.no - 5, na = lv.na;
  const bool do_bias = BWD && grad_bias != nullptr;
  constexpr bool focal_cls = FOCAL;  // focal modulation of the classification BCE (template: see yolo_obj_fwd_kernel)
  if (do_bias) {
    for (int i = threadIdx.x; i < na * no; i += blockDim.x) s_gb[i] = 0.0f;
    __syncthreads();
  }
  const long long s0 = ((long long)blockIdx.x * 4 + wave) * kSlotsPerWave;
  const long long s = s0 + lane;
  const long long so = lv.so[l];
  const float st = lv.stride[l];
  int32_t my_cell = -1, my_cls = 0;
  float4 my_tb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane < kSlotsPerWave && s < S) {
    my_cell = slot_cell[(long long)l * S + s];
    if (my_cell >= 0) {
      my_cls = slot_cls[(long long)l * S + s];
      my_tb = slot_tbox[(long long)l * S + s];
    }
  }
  unsigned long long live = __ballot(my_cell >= 0);
  float gbox = 0.0f, gcls = 0.0f;
  if (BWD) {
    const float n = acc[l * 4 + 1];
    if (n > 0.0f) {
      const float gs = grad_scale[0];
      gbox = gs * (float)B * cfg.box_gain / n;
      gcls = nc > 1 ? gs * (float)B * cfg.cls_gain / (n * (float)nc) : 0.0f;
    }
  }
  float sum_box = 0.0f, sum_cls = 0.0f, cnt = 0.0f;
  while (live) {
    const int j = __ffsll((long long)live) - 1;
    live &= live - 1;
    const int cell = __shfl(my_cell, j, 64);
    const int tc = __shfl(my_cls, j, 64);
    float4 tb;
    tb.x = __shfl(my_tb.x, j, 64);
    tb.y = __shfl(my_tb.y, j, 64);
    tb.z = __shfl(my_tb.z, j, 64);
    tb.w = __shfl(my_tb.w, j, 64);
    const CellPos cp = cell_decompose(lv, l, cell);
    const float* p = lv.pred[l] + cell_offset(lv, l, cp);
    const float aw = lv.anchor_w[l][cp.a] / st, ah = lv.anchor_h[l][cp.a] / st;
    // every lane evaluates the (wave-uniform) box term; lanes then split the class channels
    float sg[4];
    const float4 pb = yolo_pbox(p[0], p[so], p[2 * so], p[3 * so], aw, ah, sg);
    const float4 tx = tbox_xyxy(tb);
    if (!BWD) {
      const float ci = hd_ciou<false>(pb, tx, nullptr);
      float csum = 0.0f;
      if (nc > 1) {
        if (focal_cls) {
          float unused;
          for (int c = lane; c < nc; c += 64)
            csum += hd_focal_bce(p[(5 + c) * so], c == tc ? 1.0f : 0.0f, cfg.focal_gamma, cfg.focal_alpha, &unused);
        } else {
          for (int c = lane; c < nc; c += 64) csum += hd_bce_logits(p[(5 + c) * so], c == tc ? 1.0f : 0.0f);
        }
      }
      csum = hd_wave_sum(csum);
      sum_box += 1.0f - ci;
      sum_cls += csum;
      cnt += 1.0f;
      if (lane == 0) {
        const float cl = ci > 0.0f ? ci : 0.0f;
        const int ti = tobj_index(lv, l, fl.pred_dense[l], cp, cell);
        atomicMax(reinterpret_cast<int*>(tobj + ix.first_cell[l] + ti), __float_as_int(cl));
      }
    } else {
      float gc[4];
      hd_ciou<true>(pb, tx, gc);
      float* g = glv.pred[l];                       // element type G::elem behind the declared float pointer
      const long long g0 = cell_offset(glv, l, cp);
      const long long gso = glv.so[l];
      if (lane < 4) {
        const float gpx = -(gc[0] + gc[2]), gpy = -(gc[1] + gc[3]);
        const float gpw = -(gc[2] - gc[0]) * 0.5f, gph = -(gc[3] - gc[1]) * 0.5f;
        float v;
        if (lane == 0) v = gbox * gpx * 2.0f * sg[0] * (1.0f - sg[0]);
        else if (lane == 1) v = gbox * gpy * 2.0f * sg[1] * (1.0f - sg[1]);
        else if (lane == 2) v = gbox * gpw * 8.0f * sg[2] * sg[2] * (1.0f - sg[2]) * aw;
        else v = gbox * gph * 8.0f * sg[3] * sg[3] * (1.0f - sg[3]) * ah;
        G::add(g, g0 + lane * gso, v);
        if (do_bias) atomicAdd(&s_gb[cp.a * no + lane], v);
      }
      if (nc > 1)
        for (int c = lane; c < nc; c += 64) {
          float d;
          if (focal_cls) hd_focal_bce(p[(5 + c) * so], c == tc ? 1.0f : 0.0f, cfg.focal_gamma, cfg.focal_alpha, &d);
          else d = hd_sigmoid(p[(5 + c) * so]) - (c == tc ? 1.0f : 0.0f);
          const float v = gcls * d;
          G::add(g, g0 + (5 + c) * gso, v);
          if (do_bias) atomicAdd(&s_gb[cp.a * no + 5 + c], v);
        }
    }
  }
  if (!BWD) {
    // workgroup-level fold in LDS, then one atomic triple per workgroup into its replica
    __shared__ float s_red[3];
    if (threadIdx.x < 3) s_red[threadIdx.x] = 0.0f;
    __syncthreads();
    if (lane == 0 && cnt > 0.0f) {
      atomicAdd(&s_red[0], sum_box);
      atomicAdd(&s_red[1], cnt);
      atomicAdd(&s_red[2], sum_cls);
    }
    __syncthreads();
    if (threadIdx.x < 3 && s_red[1] > 0.0f) atomicAdd(acc_replica(acc, blockIdx.x) + l * 4 + threadIdx.x, s_red[threadIdx.x]);
  }
  if (do_bias) {
    // workgroup 0 of each level folds in the dense kernel's per-workgroup objectness partials
    if (blockIdx.x == 0 && obj_part != nullptr) {
      for (int a = 0; a < na; ++a) {
        float v = 0.0f;
        for (int k = threadIdx.x; k < n_part; k += blockDim.x) v += obj_part[((size_t)l * n_part + k) * na + a];
        v = hd_wave_sum(v);
        if (lane == 0) atomicAdd(&s_gb[a * no + 4], v);
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < na * no; i += blockDim.x) {
      const float v = s_gb[i];
      if (v != 0.0f) atomicAdd(grad_bias + (long long)l * na * no + i, v);
    }
  }
}

// objectness BCE over every cell of one level; grid (blocks, n_levels)
// It also overwrites tobj[m] with d(term)/d(logit) (sigmoid(x) - tobj[m] for plain BCE): backward then needs
// only that linear array (4 B per cell) instead of re-gathering the logits at a 4*no-byte stride
// (PMC: the gather fetched 28 MB raw / 56 MB corrected per launch for 3.2 MB of useful bytes).
template <bool FOCAL>  // a template, not a branch: the powf path costs registers the plain-BCE gather cannot afford
__global__ __launch_bounds__(256) void yolo_obj_fwd_kernel(const hd_yolo_levels lv, const LevelIndex ix,
                                                           const LevelFlags fl, int B, const hd_yolo_loss_cfg cfg,
                                                           float* __restrict__ tobj, float* __restrict__ acc) {
  __shared__ float s_part[4];
  const int l = blockIdx.y;
  const int cells = (int)(ix.first_cell[l + 1] - ix.first_cell[l]);
  float* to = tobj + ix.first_cell[l];
  const float* pr = lv.pred[l];
  float sum = 0.0f;
  // d(loss term)/d(logit) replaces the target in place: sigmoid(x) - t, or the focal derivative
  auto term = [&](float x, float t, float* d) -> float {
    if (FOCAL) return hd_focal_bce(x, t, cfg.focal_gamma, cfg.focal_alpha, d);
    *d = hd_sigmoid(x) - t;
    return hd_bce_logits(x, t);
  };
  if (fl.pred_dense[l]) {
    // the objectness logit is 4 bytes of every 4*no: a pure-latency gather, so each lane keeps 4
    // independent (logit, target) pairs in flight
    const int no = lv.no;
    const int step = gridDim.x * blockDim.x;
    const bool padded = fl.pred_dense[l] == 2;
    const unsigned na = (unsigned)lv.na;
    const long long pitch = lv.sx[l];
    auto logit = [&](int m) -> float {
      if (!padded) return pr[(long long)m * no + 4];
      const unsigned pix = (unsigned)m / na, a = (unsigned)m - pix * na;
      return pr[pix * pitch + a * no + 4];
    };
    int m = blockIdx.x * blockDim.x + threadIdx.x;
    for (; m + 3 * step < cells; m += 4 * step) {
      float x[4], t[4], d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        x[u] = logit(m + u * step);
        t[u] = to[m + u * step];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        sum += term(x[u], t[u], &d[u]);
        to[m + u * step] = d[u];
      }
    }
    for (; m < cells; m += step) {
      float d;
      sum += term(logit(m), to[m], &d);
      to[m] = d;
    }
  } else {
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += gridDim.x * blockDim.x) {
      const CellPos cp = cell_decompose(lv, l, c);
      float d;
      sum += term(pr[cell_offset(lv, l, cp) + 4 * lv.so[l]], to[c], &d);
      to[c] = d;
    }
  }
  sum = hd_wave_sum(sum);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicAdd(acc_replica(acc, blockIdx.x) + l * 4 + 3, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

// one 64-lane workgroup: lanes 0..31 fold the replicas into the totals (which backward reads), lane 0 finishes
__global__ void yolo_loss_finalize_kernel(const hd_yolo_levels lv, const LevelIndex ix, int B,
                                          const hd_yolo_loss_cfg cfg, float* __restrict__ acc,
                                          float* __restrict__ loss_out) {
  if (blockIdx.x != 0) return;
  if (threadIdx.x < kAccStride) {
    float v = 0.0f;
    for (int r = 0; r < kAccRep; ++r) v += acc[64 + r * kAccStride + threadIdx.x];
    acc[threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const int nc = lv.no - 5;
  float lbox = 0.0f, lobj = 0.0f, lcls = 0.0f;
  for (int l = 0; l < lv.n_levels; ++l) {
    const float n = acc[l * 4 + 1];
    const float cells = (float)(ix.first_cell[l + 1] - ix.first_cell[l]);
    if (n > 0.0f) {
      lbox += acc[l * 4 + 0] / n;
      if (nc > 1) lcls += acc[l * 4 + 2] / (n * (float)nc);
    }
    lobj += (acc[l * 4 + 3] / cells) * cfg.balance[l];
  }
  const float fb = cfg.box_gain * lbox, fo = cfg.obj_gain * lobj, fc = cfg.cls_gain * lcls;
  loss_out[0] = (float)B * (fb + fo + fc);
  loss_out[1] = fb;
  loss_out[2] = fo;
  loss_out[3] = fc;
}

// Dense gradient of every level in ONE launch (grid.y = level): zeros everywhere except the
// objectness channel.  Fast path (gradient AND prediction dense channels-last): the tensor is cut
// into units of 4 cells = `no` 16-byte words; a lane produces one word, so a wave stores 1 KiB
// contiguous.  Channel 4 of cell j of a unit falls in word (j*no+4)>>2, component (j*no+4)&3 — the
// only lanes that load anything (objectness logit + tobj, both linear in the cell index).
// All index arithmetic is 32-bit; there is one unsigned division per word.
__global__ __launch_bounds__(256) void yolo_grad_dense_kernel(const hd_yolo_levels lv, const hd_yolo_levels glv,
                                                              const LevelIndex ix, const LevelFlags fl, int B,
                                                              const hd_yolo_loss_cfg cfg,
                                                              const float* __restrict__ grad_scale,
                                                              const float* __restrict__ tobj,
                                                              float* __restrict__ grad_bias /* partials */) {
  __shared__ float s_gb[HD_MAX_ANCHORS];
  const int l = blockIdx.y;
  const int no = lv.no, na = lv.na;
  const uint32_t cells = (uint32_t)(ix.first_cell[l + 1] - ix.first_cell[l]);
  const float gobj = grad_scale[0] * (float)B * cfg.obj_gain * cfg.balance[l] / (float)cells;
  const float* to = tobj + ix.first_cell[l];
  float* g = glv.pred[l];
  if (threadIdx.x < HD_MAX_ANCHORS) s_gb[threadIdx.x] = 0.0f;
  __syncthreads();
  if (fl.grad_dense[l] == 2 && fl.pred_dense[l]) {
    // padded pixel rows: a lane produces one 16-byte word of a row of `pitch` floats; the words that hold an objectness
    // channel (a * no + 4, one per anchor) load tobj's derivative, everything else — the padding included — is zeros
    const uint32_t pitch = (uint32_t)glv.sx[l], wpp = pitch >> 2;
    const uint32_t pixels = cells / (uint32_t)na, words = pixels * wpp;
    const bool pow2 = (wpp & (wpp - 1u)) == 0u;
    const uint32_t sh = (uint32_t)__builtin_ctz(wpp);
    uint32_t wa[4];  // word of a pixel row that holds anchor a's objectness channel (wave-uniform); more anchors: the loop
#pragma unroll
    for (int a = 0; a < 4; ++a) wa[a] = a < na ? (uint32_t)(a * no + 4) >> 2 : 0xffffffffu;
    float4* g4 = reinterpret_cast<float4*>(g);
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < words; q += gridDim.x * blockDim.x) {
      const uint32_t pix = pow2 ? q >> sh : q / wpp;
      const uint32_t t = q - pix * wpp;
      int a = -1;
#pragma unroll
      for (int a2 = 0; a2 < 4; ++a2) a = t == wa[a2] ? a2 : a;
      for (int a2 = 4; a2 < na; ++a2) a = t == ((uint32_t)(a2 * no + 4) >> 2) ? a2 : a;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a >= 0) {
        const float v = gobj * to[pix * (uint32_t)na + (uint32_t)a];
        const uint32_t k = (uint32_t)(a * no + 4) & 3u;
        r.x = k == 0u ? v : 0.f;
        r.y = k == 1u ? v : 0.f;
        r.z = k == 2u ? v : 0.f;
        r.w = k == 3u ? v : 0.f;
        if (grad_bias) atomicAdd(&s_gb[a], v);
      }
      g4[q] = r;
    }
  } else if (fl.grad_dense[l] == 1 && fl.pred_dense[l]) {
    const uint32_t units = cells >> 2;
    const uint32_t words = units * (uint32_t)no;
    float4* g4 = reinterpret_cast<float4*>(g);
    const uint32_t w1 = (uint32_t)(no + 4) >> 2, w2 = (uint32_t)(2 * no + 4) >> 2, w3 = (uint32_t)(3 * no + 4) >> 2;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < words; q += gridDim.x * blockDim.x) {
      const uint32_t u = q / (uint32_t)no;
      const uint32_t t = q - u * (uint32_t)no;
      float r[4] = {0.f, 0.f, 0.f, 0.f};
      int j = -1;
      if (t == 1u) j = 0;
      else if (t == w1) j = 1;
      else if (t == w2) j = 2;
      else if (t == w3) j = 3;
      if (j >= 0) {
        const uint32_t m = u * 4u + (uint32_t)j;
        const float v = gobj * to[m];  // to[] holds sigmoid(x) - tobj since the forward pass
        r[(j * no + 4) & 3] = v;
        if (grad_bias) atomicAdd(&s_gb[m % (uint32_t)na], v);
      }
      g4[q] = make_float4(r[0], r[1], r[2], r[3]);
    }
    // cells not covered by whole units (cells % 4, at most 3): scalar tail by the first workgroup
    if (blockIdx.x == 0) {
      const uint32_t tail = (cells - units * 4u) * (uint32_t)no;
      for (uint32_t i = threadIdx.x; i < tail; i += blockDim.x) {
        const uint32_t m = units * 4u + i / (uint32_t)no;
        const uint32_t c = i % (uint32_t)no;
        float v = 0.0f;
        if (c == 4u) {
          v = gobj * to[m];
          if (grad_bias) atomicAdd(&s_gb[m % (uint32_t)na], v);
        }
        g[(size_t)m * no + c] = v;
      }
    }
  } else {
    const long long total = (long long)cells * no;
    for (long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x; f < total;
         f += (long long)gridDim.x * blockDim.x) {
      const int cell = (int)(f / no);
      const int c = (int)(f - (long long)cell * no);
      const CellPos cp = cell_decompose(lv, l, cell);
      float v = 0.0f;
      if (c == 4) {
        const int ti = tobj_index(lv, l, fl.pred_dense[l], cp, cell);
        v = gobj * to[ti];
        if (grad_bias) atomicAdd(&s_gb[cp.a], v);
      }
      g[cell_offset(glv, l, cp) + c * glv.so[l]] = v;
    }
  }
  // per-workgroup partial of the objectness bias gradient: plain store, reduced by yolo_pos_kernel<true>
  // (no zero-init, and no same-address global atomics: 3 x 4096 of those serialise to ~65 us)
  if (grad_bias) {
    __syncthreads();
    if (threadIdx.x < na) grad_bias[((size_t)l * gridDim.x + blockIdx.x) * na + threadIdx.x] = s_gb[threadIdx.x];
  }
}

// The same for a bfloat16 gradient in padded pixel rows (pitch % 8 == 0): a lane produces one 16-byte word = 8 channels.
__global__ __launch_bounds__(256) void yolo_grad_dense_bf16_kernel(const hd_yolo_levels lv, const hd_yolo_levels glv,
                                                                   const LevelIndex ix, int B, const hd_yolo_loss_cfg cfg,
                                                                   const float* __restrict__ grad_scale,
                                                                   const float* __restrict__ tobj,
                                                                   float* __restrict__ grad_bias /* partials */) {
  __shared__ float s_gb[HD_MAX_ANCHORS];
  const int l = blockIdx.y;
  const int no = lv.no, na = lv.na;
  const uint32_t cells = (uint32_t)(ix.first_cell[l + 1] - ix.first_cell[l]);
  const float gobj = grad_scale[0] * (float)B * cfg.obj_gain * cfg.balance[l] / (float)cells;
  const float* to = tobj + ix.first_cell[l];
  if (threadIdx.x < HD_MAX_ANCHORS) s_gb[threadIdx.x] = 0.0f;
  __syncthreads();
  const uint32_t pitch = (uint32_t)glv.sx[l], wpp = pitch >> 3;
  const uint32_t pixels = cells / (uint32_t)na, words = pixels * wpp;
  const bool pow2 = (wpp & (wpp - 1u)) == 0u;
  const uint32_t sh = (uint32_t)__builtin_ctz(wpp);
  uint32_t wa[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) wa[a] = a < na ? (uint32_t)(a * no + 4) >> 3 : 0xffffffffu;
  uint4* g4 = reinterpret_cast<uint4*>(glv.pred[l]);
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < words; q += gridDim.x * blockDim.x) {
    const uint32_t pix = pow2 ? q >> sh : q / wpp;
    const uint32_t t = q - pix * wpp;
    // every anchor whose objectness channel a * no + 4 falls into this 8-channel word: with no < 8 (one or two classes) two
    // anchors can share a word (no = 6: channels 16 and 22), so all of them are folded in, not just the last match
    uint32_t r4[4] = {0u, 0u, 0u, 0u};
    auto put = [&](int a2) {
      const float v = gobj * to[pix * (uint32_t)na + (uint32_t)a2];
      const uint32_t k = (uint32_t)(a2 * no + 4) & 7u;
      const uint32_t h = (uint32_t)GradBf16::round_bf16(v) << ((k & 1u) * 16u);
      r4[0] |= (k >> 1) == 0u ? h : 0u;
      r4[1] |= (k >> 1) == 1u ? h : 0u;
      r4[2] |= (k >> 1) == 2u ? h : 0u;
      r4[3] |= (k >> 1) == 3u ? h : 0u;
      if (grad_bias) atomicAdd(&s_gb[a2], v);   // the bias gradient sums the unrounded terms
    };
#pragma unroll
    for (int a2 = 0; a2 < 4; ++a2)
      if (t == wa[a2]) put(a2);
    for (int a2 = 4; a2 < na; ++a2)
      if (t == ((uint32_t)(a2 * no + 4) >> 3)) put(a2);
    const uint4 r = make_uint4(r4[0], r4[1], r4[2], r4[3]);
    g4[q] = r;
  }
  if (grad_bias) {
    __syncthreads();
    if (threadIdx.x < na) grad_bias[((size_t)l * gridDim.x + blockIdx.x) * na + threadIdx.x] = s_gb[threadIdx.x];
  }
}

constexpr int kDenseMaxBlocks = 4096;  // 16 workgroups per CU per level, grid-stride beyond

// Workgroups per level of the dense-gradient launch: 256 work items per workgroup sweep.
inline unsigned dense_grid_x(const hd_yolo_levels& lv, const hd_yolo_levels& glv, const LevelFlags& fl, const LevelIndex& ix,
                             int grad_dtype = 0) {
  long long max_items = 0;
  for (int l = 0; l < lv.n_levels; ++l) {
    const long long cells = ix.first_cell[l + 1] - ix.first_cell[l];
    const long long total = cells * lv.no;
    const bool fast = (grad_dtype == 1 || fl.grad_dense[l]) && fl.pred_dense[l];
    const long long words = grad_dtype == 1 ? cells / lv.na * (glv.sx[l] / 8)
                            : fl.grad_dense[l] == 2 ? cells / lv.na * (glv.sx[l] / 4) : (total + 3) / 4;
    max_items = std::max(max_items, fast ? words : total);
  }
  long long gx = (max_items + 255) / 256;
  if (gx > kDenseMaxBlocks) gx = kDenseMaxBlocks;
  if (gx < 1) gx = 1;
  return (unsigned)gx;
}

inline unsigned cap_grid(long long work_items, int per_block) {
  long long b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (unsigned)b;
}

inline LevelFlags make_flags(const hd_yolo_levels& lv, const hd_yolo_levels* glv, int B) {
  LevelFlags f;
  for (int l = 0; l < HD_MAX_LEVELS; ++l) {
    f.pred_dense[l] = l < lv.n_levels ? is_dense_nhwc(lv, l, B) : 0;
    f.grad_dense[l] = (glv && l < lv.n_levels) ? is_dense_nhwc(*glv, l, B) : 0;
  }
  return f;
}

}  // namespace

extern "C" hd_status hd_yolo_decode(const hd_yolo_levels* lv, int B, float* out, hd_stream stream) {
  hd_status st = check_levels(lv, B);
  if (st != HD_OK) return st;
  if (!out) return HD_ERR_NULL;
  if (!hd_aligned16(out)) return HD_ERR_ALIGN;
  const LevelIndex ix = make_index(*lv, B);
  bool all_dense = true;
  for (int l = 0; l < lv->n_levels; ++l) all_dense = all_dense && is_dense_nhwc_decl(*lv, l, B);
  if (all_dense && (size_t)lv->na * kDenseCells * lv->no * sizeof(float) <= 64 * 1024) {
    DecodePlan dplan;
    long long dblocks = 0;
    for (int l = 0; l <= HD_MAX_LEVELS; ++l) {
      dplan.first_block[l] = (int)dblocks;
      if (l < lv->n_levels) {
        dplan.tiles[l] = (lv->h[l] * lv->w[l] + kDenseCells - 1) / kDenseCells;
        dblocks += (long long)dplan.tiles[l] * B;
      } else if (l < HD_MAX_LEVELS) {
        dplan.tiles[l] = 1;
      }
    }
    if (dblocks > 0x7fffffffLL) return HD_ERR_SHAPE;
    const size_t dlds = (size_t)lv->na * kDenseCells * lv->no * sizeof(float);
    hipLaunchKernelGGL(yolo_decode_dense_kernel, dim3((unsigned)dblocks), dim3(256), dlds, (hipStream_t)stream, *lv,
                       ix, dplan, B, out);
    HD_RETURN_IF_LAUNCH_FAILED();
    return HD_OK;
  }
  DecodePlan plan;
  long long blocks = 0;
  for (int l = 0; l <= HD_MAX_LEVELS; ++l) {
    plan.first_block[l] = (int)blocks;
    if (l < lv->n_levels) {
      plan.tiles[l] = (lv->h[l] * lv->w[l] + kDecRows - 1) / kDecRows;
      blocks += (long long)plan.tiles[l] * lv->na * B;
    } else if (l < HD_MAX_LEVELS) {
      plan.tiles[l] = 1;
    }
  }
  if (blocks > 0x7fffffffLL) return HD_ERR_SHAPE;
  const size_t lds = ((size_t)kDecRows * lv->no + 1) * sizeof(float) + kDecRows * (sizeof(long long) + 2 * sizeof(int));
  if (lds > 64 * 1024) return HD_ERR_SHAPE;  // no > ~250 channels would need a smaller row tile
  hipLaunchKernelGGL(yolo_decode_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, *lv, ix, plan,
                     B, out);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_yolo_assign(const hd_yolo_levels* lv, const float* gt_boxes, const int32_t* gt_cls,
                                    const int32_t* m_valid, int B, int M, float anchor_t, int32_t* slot_cell,
                                    float* slot_tbox, int32_t* slot_cls, hd_stream stream) {
  if (!lv) return HD_ERR_NULL;
  if (lv->n_levels <= 0 || lv->n_levels > HD_MAX_LEVELS || lv->na <= 0 || lv->na > HD_MAX_ANCHORS || B <= 0 || M < 0)
    return HD_ERR_SHAPE;
  if (M == 0) return HD_OK;
  if (!gt_boxes || !gt_cls || !m_valid || !slot_cell || !slot_tbox || !slot_cls) return HD_ERR_NULL;
  if (!hd_aligned16(gt_boxes) || !hd_aligned16(slot_tbox)) return HD_ERR_ALIGN;
  for (int l = 0; l < lv->n_levels; ++l)
    if ((long long)B * lv->na * lv->h[l] * lv->w[l] > 0x7fffffffLL || lv->h[l] <= 0 || lv->w[l] <= 0)
      return HD_ERR_SHAPE;
  const long long total = (long long)5 * lv->na * B * M * lv->n_levels;
  hipLaunchKernelGGL(yolo_assign_kernel, dim3(cap_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, *lv,
                     reinterpret_cast<const float4*>(gt_boxes), gt_cls, m_valid, B, M, anchor_t, slot_cell,
                     reinterpret_cast<float4*>(slot_tbox), slot_cls);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

// workspace = [acc: 256 B][bias-gradient partials: n_levels * kDenseMaxBlocks * na floats]
//             [tobj: total cells * 4 B]
static constexpr size_t kAccBytes = 256 + (size_t)kAccRep * kAccStride * sizeof(float);  // totals + replicas

static size_t part_bytes(const hd_yolo_levels* lv) {
  return hd_align_up((size_t)lv->n_levels * kDenseMaxBlocks * lv->na * sizeof(float), 256);
}

extern "C" size_t hd_yolo_loss_workspace_bytes(const hd_yolo_levels* lv, int B) {
  if (!lv || lv->n_levels <= 0 || lv->n_levels > HD_MAX_LEVELS || B <= 0 || lv->na <= 0) return 0;
  const LevelIndex ix = make_index(*lv, B);
  return kAccBytes + part_bytes(lv) + hd_align_up((size_t)ix.first_cell[lv->n_levels] * sizeof(float), 256);
}

static dim3 pos_grid(long long S, int n_levels) {
  const long long per_block = 4 * kSlotsPerWave;
  return dim3((unsigned)((S + per_block - 1) / per_block), (unsigned)n_levels);
}

extern "C" hd_status hd_yolo_loss_fwd(const hd_yolo_levels* lv, int B, int M, const int32_t* slot_cell,
                                      const float* slot_tbox, const int32_t* slot_cls,
                                      const hd_yolo_loss_cfg* cfg, float* loss_out, void* workspace,
                                      size_t workspace_bytes, hd_stream stream) {
  hd_status st = check_levels(lv, B);
  if (st != HD_OK) return st;
  if (!cfg || !loss_out || (M > 0 && (!slot_cell || !slot_tbox || !slot_cls))) return HD_ERR_NULL;
  if (M < 0) return HD_ERR_SHAPE;
  const size_t need = hd_yolo_loss_workspace_bytes(lv, B);
  if (!workspace || workspace_bytes < need) return HD_ERR_WORKSPACE;
  if (!hd_aligned16(workspace) || (M > 0 && !hd_aligned16(slot_tbox))) return HD_ERR_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  const LevelIndex ix = make_index(*lv, B);
  const LevelFlags fl = make_flags(*lv, nullptr, B);
  float* acc = reinterpret_cast<float*>(workspace);
  float* tobj = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + kAccBytes + part_bytes(lv));
  // acc and tobj are adjacent except for the partials area: one fill over [acc .. tobj end) would
  // also clear the (large, unused-in-forward) partials, so fill the two pieces separately
  if (hd_fill32(acc, 0u, kAccBytes / 4, s) != HD_OK) return HD_ERR_LAUNCH;
  if (hd_fill32(tobj, 0u, (size_t)ix.first_cell[lv->n_levels], s) != HD_OK) return HD_ERR_LAUNCH;
  const long long S = (long long)5 * lv->na * B * M;
  if (S > 0) {
    if (cfg->focal_gamma > 0.0f && (cfg->focal_terms & 1))
      hipLaunchKernelGGL((yolo_pos_kernel<false, true>), pos_grid(S, lv->n_levels), dim3(256), 0, s, *lv, *lv, ix, fl,
                         B, S, slot_cell, reinterpret_cast<const float4*>(slot_tbox), slot_cls, *cfg,
                         (const float*)nullptr, acc, tobj, (float*)nullptr, (const float*)nullptr, 0);
    else
      hipLaunchKernelGGL((yolo_pos_kernel<false, false>), pos_grid(S, lv->n_levels), dim3(256), 0, s, *lv, *lv, ix, fl,
                         B, S, slot_cell, reinterpret_cast<const float4*>(slot_tbox), slot_cls, *cfg,
                         (const float*)nullptr, acc, tobj, (float*)nullptr, (const float*)nullptr, 0);
    HD_RETURN_IF_LAUNCH_FAILED();
  }
  long long max_cells = 0;
  for (int l = 0; l < lv->n_levels; ++l) max_cells = std::max(max_cells, ix.first_cell[l + 1] - ix.first_cell[l]);
  dim3 ogrid(cap_grid(max_cells, 256 * 4), (unsigned)lv->n_levels);
  if (cfg->focal_gamma > 0.0f && (cfg->focal_terms & 2))
    hipLaunchKernelGGL(yolo_obj_fwd_kernel<true>, ogrid, dim3(256), 0, s, *lv, ix, fl, B, *cfg, tobj, acc);
  else
    hipLaunchKernelGGL(yolo_obj_fwd_kernel<false>, ogrid, dim3(256), 0, s, *lv, ix, fl, B, *cfg, tobj, acc);
  HD_RETURN_IF_LAUNCH_FAILED();
  hipLaunchKernelGGL(yolo_loss_finalize_kernel, dim3(1), dim3(64), 0, s, *lv, ix, B, *cfg, acc, loss_out);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

static hd_status bwd_common_checks(const hd_yolo_levels* lv, const hd_yolo_levels* grad_lv, int B,
                                   const hd_yolo_loss_cfg* cfg, const float* grad_scale, const void* workspace,
                                   size_t workspace_bytes) {
  hd_status st = check_levels(lv, B);
  if (st != HD_OK) return st;
  st = check_levels(grad_lv, B);
  if (st != HD_OK) return st;
  if (!cfg || !grad_scale) return HD_ERR_NULL;
  if (grad_lv->n_levels != lv->n_levels || grad_lv->na != lv->na || grad_lv->no != lv->no) return HD_ERR_SHAPE;
  for (int l = 0; l < lv->n_levels; ++l)
    if (grad_lv->h[l] != lv->h[l] || grad_lv->w[l] != lv->w[l]) return HD_ERR_SHAPE;
  if (!workspace || workspace_bytes < hd_yolo_loss_workspace_bytes(lv, B)) return HD_ERR_WORKSPACE;
  return HD_OK;
}

// grad_dtype 1: bfloat16 gradient elements in padded pixel rows (so == 1, sa == no, sx % 8 == 0, rows contiguous, 16-byte
// aligned), predictions dense (either kind) so that tobj is in pixel order — what ops.head_conv produces.
static hd_status bf16_grad_checks(const hd_yolo_levels* lv, const hd_yolo_levels* g, const LevelFlags& fl, int B) {
  for (int l = 0; l < lv->n_levels; ++l) {
    const long long no = g->no, na = g->na, W = g->w[l], H = g->h[l], P = g->sx[l];
    if (!fl.pred_dense[l]) return HD_ERR_SHAPE;
    if (!(g->so[l] == 1 && g->sa[l] == no && P >= na * no && P % 8 == 0 && g->sy[l] == W * P && (B == 1 || g->sb[l] == H * W * P)))
      return HD_ERR_SHAPE;
    if (!hd_aligned16(g->pred[l])) return HD_ERR_ALIGN;
  }
  return HD_OK;
}

static hd_status bwd_dense_impl(const hd_yolo_levels* lv, const hd_yolo_levels* grad_lv, int B, const hd_yolo_loss_cfg* cfg,
                                const float* grad_scale, float* grad_bias, const void* workspace, size_t workspace_bytes,
                                hd_stream stream, int grad_dtype) {
  hd_status st = bwd_common_checks(lv, grad_lv, B, cfg, grad_scale, workspace, workspace_bytes);
  if (st != HD_OK) return st;
  hipStream_t s = (hipStream_t)stream;
  const LevelIndex ix = make_index(*lv, B);
  const LevelFlags fl = make_flags(*lv, grad_dtype ? nullptr : grad_lv, B);
  if (grad_dtype && (st = bf16_grad_checks(lv, grad_lv, fl, B)) != HD_OK) return st;
  const float* tobj =
      reinterpret_cast<const float*>(reinterpret_cast<const char*>(workspace) + kAccBytes + part_bytes(lv));
  float* part = grad_bias ? reinterpret_cast<float*>(reinterpret_cast<char*>(const_cast<void*>(workspace)) + kAccBytes)
                          : nullptr;
  const unsigned gx = dense_grid_x(*lv, *grad_lv, fl, ix, grad_dtype);
  if (grad_dtype)
    (void)hd_launch(yolo_grad_dense_bf16_kernel, dim3(gx, (unsigned)lv->n_levels), dim3(256), 0, s, *lv, *grad_lv, ix, B, *cfg,
                    grad_scale, tobj, part);
  else
    (void)hd_launch(yolo_grad_dense_kernel, dim3(gx, (unsigned)lv->n_levels), dim3(256), 0, s, *lv, *grad_lv, ix, fl,
                    B, *cfg, grad_scale, tobj, part);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_yolo_loss_bwd_dense(const hd_yolo_levels* lv, const hd_yolo_levels* grad_lv, int B,
                                            const hd_yolo_loss_cfg* cfg, const float* grad_scale,
                                            float* grad_bias, const void* workspace, size_t workspace_bytes,
                                            hd_stream stream) {
  return bwd_dense_impl(lv, grad_lv, B, cfg, grad_scale, grad_bias, workspace, workspace_bytes, stream, 0);
}

extern "C" hd_status hd_yolo_loss_bwd_dense_bf16(const hd_yolo_levels* lv, const hd_yolo_levels* grad_lv, int B,
                                                 const hd_yolo_loss_cfg* cfg, const float* grad_scale, float* grad_bias,
                                                 const void* workspace, size_t workspace_bytes, hd_stream stream) {
  return bwd_dense_impl(lv, grad_lv, B, cfg, grad_scale, grad_bias, workspace, workspace_bytes, stream, 1);
}

static hd_status bwd_pos_impl(const hd_yolo_levels* lv, const hd_yolo_levels* grad_lv, int B, int M,
                              const int32_t* slot_cell, const float* slot_tbox, const int32_t* slot_cls,
                              const hd_yolo_loss_cfg* cfg, const float* grad_scale, float* grad_bias,
                              const void* workspace, size_t workspace_bytes, hd_stream stream, int grad_dtype) {
  hd_status st = bwd_common_checks(lv, grad_lv, B, cfg, grad_scale, workspace, workspace_bytes);
  if (st != HD_OK) return st;
  if (M < 0) return HD_ERR_SHAPE;
  if (M == 0 && !grad_bias) return HD_OK;
  if (M > 0 && (!slot_cell || !slot_tbox || !slot_cls)) return HD_ERR_NULL;
  const LevelIndex ix = make_index(*lv, B);
  const LevelFlags fl = make_flags(*lv, grad_dtype ? nullptr : grad_lv, B);
  if (grad_dtype && (st = bf16_grad_checks(lv, grad_lv, fl, B)) != HD_OK) return st;
  char* wsb = reinterpret_cast<char*>(const_cast<void*>(workspace));
  float* acc = reinterpret_cast<float*>(wsb);
  const float* part = reinterpret_cast<const float*>(wsb + kAccBytes);
  float* tobj = reinterpret_cast<float*>(wsb + kAccBytes + part_bytes(lv));
  const long long S = (long long)5 * lv->na * B * M;
  const size_t lds = grad_bias ? (size_t)lv->na * lv->no * sizeof(float) : 0;
  dim3 grid = S > 0 ? pos_grid(S, lv->n_levels) : dim3(1, (unsigned)lv->n_levels);  // M == 0: partial reduce only
  const bool focal = cfg->focal_gamma > 0.0f && (cfg->focal_terms & 1);
  const int n_part = (int)dense_grid_x(*lv, *grad_lv, fl, ix, grad_dtype);
#define HD_POS(FOCAL, GT)                                                                                                  \
  hipLaunchKernelGGL((yolo_pos_kernel<true, FOCAL, GT>), grid, dim3(256), lds, (hipStream_t)stream, *lv, *grad_lv, ix, fl, B, S, \
                     slot_cell, reinterpret_cast<const float4*>(slot_tbox), slot_cls, *cfg, grad_scale, acc, tobj, grad_bias, \
                     grad_bias ? part : nullptr, n_part)
  if (grad_dtype) {
    if (focal) HD_POS(true, GradBf16);
    else HD_POS(false, GradBf16);
  } else {
    if (focal) HD_POS(true, GradF32);
    else HD_POS(false, GradF32);
  }
#undef HD_POS
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_yolo_loss_bwd_pos(const hd_yolo_levels* lv, const hd_yolo_levels* grad_lv, int B, int M,
                                          const int32_t* slot_cell, const float* slot_tbox, const int32_t* slot_cls,
                                          const hd_yolo_loss_cfg* cfg, const float* grad_scale, float* grad_bias,
                                          const void* workspace, size_t workspace_bytes, hd_stream stream) {
  return bwd_pos_impl(lv, grad_lv, B, M, slot_cell, slot_tbox, slot_cls, cfg, grad_scale, grad_bias, workspace, workspace_bytes,
                      stream, 0);
}

extern "C" hd_status hd_yolo_loss_bwd_pos_bf16(const hd_yolo_levels* lv, const hd_yolo_levels* grad_lv, int B, int M,
                                               const int32_t* slot_cell, const float* slot_tbox, const int32_t* slot_cls,
                                               const hd_yolo_loss_cfg* cfg, const float* grad_scale, float* grad_bias,
                                               const void* workspace, size_t workspace_bytes, hd_stream stream) {
  return bwd_pos_impl(lv, grad_lv, B, M, slot_cell, slot_tbox, slot_cls, cfg, grad_scale, grad_bias, workspace, workspace_bytes,
                      stream, 1);
}

extern "C" hd_status hd_yolo_loss_bwd(const hd_yolo_levels* lv, const hd_yolo_levels* grad_lv, int B, int M,
                                      const int32_t* slot_cell, const float* slot_tbox, const int32_t* slot_cls,
                                      const hd_yolo_loss_cfg* cfg, const float* grad_scale, float* grad_bias,
                                      const void* workspace, size_t workspace_bytes, hd_stream stream) {
  hd_status st =
      hd_yolo_loss_bwd_dense(lv, grad_lv, B, cfg, grad_scale, grad_bias, workspace, workspace_bytes, stream);
  if (st != HD_OK) return st;
  return hd_yolo_loss_bwd_pos(lv, grad_lv, B, M, slot_cell, slot_tbox, slot_cls, cfg, grad_scale, grad_bias,
                              workspace, workspace_bytes, stream);
}
