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

// ------------------------------------------------------------------------------------ decode

__global__ __launch_bounds__(256) void yolo_decode_kernel(const hd_yolo_levels lv, const LevelIndex ix, int B,
                                                          float* __restrict__ out) {
  const int no = lv.no;
  const long long n_total = ix.first_row[lv.n_levels];
  const long long total = (long long)B * n_total * no;
  for (long long f0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; f0 < total;
       f0 += (long long)gridDim.x * blockDim.x * 4) {
    float r[4];
    long long grow = f0 / no;  // global row = b*n_total + row
    int c = (int)(f0 - grow * no);
    // decode (b, level, a, y, x) of the first element, then step
    int b = (int)(grow / n_total);
    long long row = grow - (long long)b * n_total;
    int l = 0;
#pragma unroll
    for (int k = 1; k < HD_MAX_LEVELS; ++k)
      if (k < lv.n_levels && row >= ix.first_row[k]) l = k;
    long long rr = row - ix.first_row[l];
    int W = lv.w[l], H = lv.h[l];
    int x = (int)(rr % W);
    long long t = rr / W;
    int y = (int)(t % H);
    int a = (int)(t / H);
    const float* p = lv.pred[l] + b * lv.sb[l] + a * lv.sa[l] + y * lv.sy[l] + x * lv.sx[l];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (f0 + k < total) {
        const float s = hd_sigmoid(p[c * lv.so[l]]);
        float v;
        if (c == 0) v = (s * 2.0f - 0.5f + (float)x) * lv.stride[l];
        else if (c == 1) v = (s * 2.0f - 0.5f + (float)y) * lv.stride[l];
        else if (c == 2) v = (s * 2.0f) * (s * 2.0f) * lv.anchor_w[l][a];
        else if (c == 3) v = (s * 2.0f) * (s * 2.0f) * lv.anchor_h[l][a];
        else v = s;
        r[k] = v;
        if (++c == no) {  // next row: advance (x, y, a, level, b)
          c = 0;
          ++row;
          if (++x == W) {
            x = 0;
            if (++y == H) {
              y = 0;
              if (++a == lv.na) {
                a = 0;
                ++l;
                if (l == lv.n_levels) {
                  l = 0;
                  ++b;
                  row = 0;
                }
                W = lv.w[l];
                H = lv.h[l];
              }
            }
          }
          if (b < B) p = lv.pred[l] + b * lv.sb[l] + a * lv.sa[l] + y * lv.sy[l] + x * lv.sx[l];
        }
      } else {
        r[k] = 0.0f;
      }
    }
    if (f0 + 3 < total) {
      *reinterpret_cast<float4*>(out + f0) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
      for (int k = 0; k < 4; ++k)
        if (f0 + k < total) out[f0 + k] = r[k];
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

struct CellRef {
  const float* p;
  int a;
};

__device__ __forceinline__ long long cell_offset(const hd_yolo_levels& lv, int l, int cell, int* a_out) {
  const int W = lv.w[l], H = lv.h[l];
  const int gi = cell % W;
  int r = cell / W;
  const int gj = r % H;
  r /= H;
  const int a = r % lv.na;
  const int b = r / lv.na;
  *a_out = a;
  return b * lv.sb[l] + a * lv.sa[l] + gj * lv.sy[l] + gi * lv.sx[l];
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

// grid: (ceil(S/256), n_levels); each wave owns 64 consecutive slots of one level
template <bool BWD>
__global__ __launch_bounds__(256) void yolo_pos_kernel(const hd_yolo_levels lv, const hd_yolo_levels glv,
                                                       const LevelIndex ix, int B, long long S,
                                                       const int32_t* __restrict__ slot_cell,
                                                       const float4* __restrict__ slot_tbox,
                                                       const int32_t* __restrict__ slot_cls,
                                                       const hd_yolo_loss_cfg cfg,
                                                       const float* __restrict__ grad_scale, float* __restrict__ acc,
                                                       float* __restrict__ tobj) {
  const int l = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int no = lv.no, nc = lv.no - 5;
  const long long so = lv.so[l];
  const float st = lv.stride[l];
  int32_t my_cell = -1, my_cls = 0;
  float4 my_tb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (s < S) {
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
    int a;
    const long long off = cell_offset(lv, l, cell, &a);
    const float* p = lv.pred[l] + off;
    const float aw = lv.anchor_w[l][a] / st, ah = lv.anchor_h[l][a] / st;
    // every lane evaluates the (wave-uniform) box term; lanes then split the class channels
    float sg[4];
    const float4 pb = yolo_pbox(p[0], p[so], p[2 * so], p[3 * so], aw, ah, sg);
    const float4 tx = tbox_xyxy(tb);
    if (!BWD) {
      const float ci = hd_ciou<false>(pb, tx, nullptr);
      float csum = 0.0f;
      if (nc > 1)
        for (int c = lane; c < nc; c += 64) csum += hd_bce_logits(p[(5 + c) * so], c == tc ? 1.0f : 0.0f);
      csum = hd_wave_sum(csum);
      sum_box += 1.0f - ci;
      sum_cls += csum;
      cnt += 1.0f;
      if (lane == 0) {
        const float cl = ci > 0.0f ? ci : 0.0f;
        atomicMax(reinterpret_cast<int*>(tobj + ix.first_cell[l] + cell), __float_as_int(cl));
      }
    } else {
      float gc[4];
      hd_ciou<true>(pb, tx, gc);
      float* g = glv.pred[l] + cell_offset(glv, l, cell, &a);
      const long long gso = glv.so[l];
      if (lane == 0) {
        const float gpx = -(gc[0] + gc[2]), gpy = -(gc[1] + gc[3]);
        const float gpw = -(gc[2] - gc[0]) * 0.5f, gph = -(gc[3] - gc[1]) * 0.5f;
        atomicAdd(g + 0 * gso, gbox * gpx * 2.0f * sg[0] * (1.0f - sg[0]));
        atomicAdd(g + 1 * gso, gbox * gpy * 2.0f * sg[1] * (1.0f - sg[1]));
        atomicAdd(g + 2 * gso, gbox * gpw * 8.0f * sg[2] * sg[2] * (1.0f - sg[2]) * aw);
        atomicAdd(g + 3 * gso, gbox * gph * 8.0f * sg[3] * sg[3] * (1.0f - sg[3]) * ah);
      }
      if (nc > 1)
        for (int c = lane; c < nc; c += 64)
          atomicAdd(g + (5 + c) * gso, gcls * (hd_sigmoid(p[(5 + c) * so]) - (c == tc ? 1.0f : 0.0f)));
    }
  }
  if (!BWD && lane == 0 && cnt > 0.0f) {
    atomicAdd(acc + l * 4 + 0, sum_box);
    atomicAdd(acc + l * 4 + 1, cnt);
    atomicAdd(acc + l * 4 + 2, sum_cls);
  }
  (void)no;
}

// objectness BCE over every cell of one level; grid (blocks, n_levels)
__global__ __launch_bounds__(256) void yolo_obj_fwd_kernel(const hd_yolo_levels lv, const LevelIndex ix, int B,
                                                           const float* __restrict__ tobj, float* __restrict__ acc) {
  __shared__ float s_part[4];
  const int l = blockIdx.y;
  const long long cells = ix.first_cell[l + 1] - ix.first_cell[l];
  const float* to = tobj + ix.first_cell[l];
  float sum = 0.0f;
  for (long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x; c < cells;
       c += (long long)gridDim.x * blockDim.x) {
    int a;
    const long long off = cell_offset(lv, l, (int)c, &a);
    sum += hd_bce_logits(lv.pred[l][off + 4 * lv.so[l]], to[c]);
  }
  sum = hd_wave_sum(sum);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float tot = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    atomicAdd(acc + l * 4 + 3, tot);
  }
}

__global__ void yolo_loss_finalize_kernel(const hd_yolo_levels lv, const LevelIndex ix, int B,
                                          const hd_yolo_loss_cfg cfg, const float* __restrict__ acc,
                                          float* __restrict__ loss_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
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

// Dense gradient of every level in ONE launch (grid.y = level).  Fast path per level when the
// gradient tensor is one contiguous block in memory order (b, y, x, a, c)  [sx = na*no, sa = no,
// so = 1, sy = W*sx, sb = H*sy]: each lane writes 4 consecutive floats (16-byte store); the only
// non-zero is channel 4 (objectness).  Other layouts: one lane per (cell, channel).
struct DenseFlags {
  int fast[HD_MAX_LEVELS];
};

__global__ __launch_bounds__(256) void yolo_grad_dense_kernel(const hd_yolo_levels lv, const hd_yolo_levels glv,
                                                              const LevelIndex ix, const DenseFlags flags, int B,
                                                              const hd_yolo_loss_cfg cfg,
                                                              const float* __restrict__ grad_scale,
                                                              const float* __restrict__ tobj) {
  const int l = blockIdx.y;
  const int no = lv.no, na = lv.na, H = lv.h[l], W = lv.w[l];
  const long long cells = (long long)B * na * H * W;
  const long long total = cells * no;
  const float gobj = grad_scale[0] * (float)B * cfg.obj_gain * cfg.balance[l] / (float)cells;
  const float* to = tobj + ix.first_cell[l];
  float* g = glv.pred[l];
  if (flags.fast[l]) {
    for (long long f0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; f0 < total;
         f0 += (long long)gridDim.x * blockDim.x * 4) {
      float r[4] = {0.f, 0.f, 0.f, 0.f};
      long long mrow = f0 / no;  // memory-order cell index: ((b*H + y)*W + x)*na + a
      int c = (int)(f0 - mrow * no);
      // the 4 elements span at most 2 cells; channel 4 occurs at most once (no >= 5)
      int k4 = 4 - c;  // offset of channel 4 of cell mrow
      if (k4 < 0) {
        k4 += no;
        ++mrow;
      }
      if (k4 < 4 && f0 + k4 < total) {
        const int a = (int)(mrow % na);
        long long t = mrow / na;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        const long long cell = (((long long)b * na + a) * H + y) * W + x;
        const float xo = lv.pred[l][b * lv.sb[l] + a * lv.sa[l] + y * lv.sy[l] + x * lv.sx[l] + 4 * lv.so[l]];
        r[k4] = gobj * (hd_sigmoid(xo) - to[cell]);
      }
      if (f0 + 3 < total) {
        *reinterpret_cast<float4*>(g + f0) = make_float4(r[0], r[1], r[2], r[3]);
      } else {
        for (int k = 0; k < 4; ++k)
          if (f0 + k < total) g[f0 + k] = r[k];
      }
    }
  } else {
    for (long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x; f < total;
         f += (long long)gridDim.x * blockDim.x) {
      const long long cell = f / no;
      const int c = (int)(f - cell * no);
      int a;
      const long long goff = cell_offset(glv, l, (int)cell, &a);
      float v = 0.0f;
      if (c == 4) {
        const long long off = cell_offset(lv, l, (int)cell, &a);
        v = gobj * (hd_sigmoid(lv.pred[l][off + 4 * lv.so[l]]) - to[cell]);
      }
      g[goff + c * glv.so[l]] = v;
    }
  }
}

inline bool is_dense_nhwc(const hd_yolo_levels& g, int l, int B) {
  const long long no = g.no, na = g.na, W = g.w[l], H = g.h[l];
  return g.so[l] == 1 && g.sa[l] == no && g.sx[l] == na * no && g.sy[l] == W * na * no &&
         (B == 1 || g.sb[l] == H * W * na * no) && hd_aligned16(g.pred[l]);
}

inline unsigned cap_grid(long long work_items, int per_block) {
  long long b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (unsigned)b;
}

}  // namespace

extern "C" hd_status hd_yolo_decode(const hd_yolo_levels* lv, int B, float* out, hd_stream stream) {
  hd_status st = check_levels(lv, B);
  if (st != HD_OK) return st;
  if (!out) return HD_ERR_NULL;
  if (!hd_aligned16(out)) return HD_ERR_ALIGN;
  const LevelIndex ix = make_index(*lv, B);
  const long long total = (long long)B * ix.first_row[lv->n_levels] * lv->no;
  hipLaunchKernelGGL(yolo_decode_kernel, dim3(cap_grid(total, 1024)), dim3(256), 0, (hipStream_t)stream, *lv, ix, B,
                     out);
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

// workspace = [acc: 256 B][tobj: total cells * 4 B]
static constexpr size_t kAccBytes = 256;

extern "C" size_t hd_yolo_loss_workspace_bytes(const hd_yolo_levels* lv, int B) {
  if (!lv || lv->n_levels <= 0 || lv->n_levels > HD_MAX_LEVELS || B <= 0) return 0;
  const LevelIndex ix = make_index(*lv, B);
  return kAccBytes + hd_align_up((size_t)ix.first_cell[lv->n_levels] * sizeof(float), 256);
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
  float* acc = reinterpret_cast<float*>(workspace);
  float* tobj = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + kAccBytes);
  if (hipMemsetAsync(workspace, 0, need, s) != hipSuccess) return HD_ERR_LAUNCH;
  const long long S = (long long)5 * lv->na * B * M;
  if (S > 0) {
    dim3 grid((unsigned)((S + 255) / 256), (unsigned)lv->n_levels);
    hipLaunchKernelGGL((yolo_pos_kernel<false>), grid, dim3(256), 0, s, *lv, *lv, ix, B, S, slot_cell,
                       reinterpret_cast<const float4*>(slot_tbox), slot_cls, *cfg, (const float*)nullptr, acc, tobj);
    HD_RETURN_IF_LAUNCH_FAILED();
  }
  long long max_cells = 0;
  for (int l = 0; l < lv->n_levels; ++l) max_cells = std::max(max_cells, ix.first_cell[l + 1] - ix.first_cell[l]);
  dim3 ogrid(cap_grid(max_cells, 256 * 4), (unsigned)lv->n_levels);
  hipLaunchKernelGGL(yolo_obj_fwd_kernel, ogrid, dim3(256), 0, s, *lv, ix, B, tobj, acc);
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

extern "C" hd_status hd_yolo_loss_bwd_dense(const hd_yolo_levels* lv, const hd_yolo_levels* grad_lv, int B,
                                            const hd_yolo_loss_cfg* cfg, const float* grad_scale,
                                            const void* workspace, size_t workspace_bytes, hd_stream stream) {
  hd_status st = bwd_common_checks(lv, grad_lv, B, cfg, grad_scale, workspace, workspace_bytes);
  if (st != HD_OK) return st;
  const LevelIndex ix = make_index(*lv, B);
  const float* tobj = reinterpret_cast<const float*>(reinterpret_cast<const char*>(workspace) + kAccBytes);
  DenseFlags flags;
  long long max_total = 0;
  bool any_fast = false;
  for (int l = 0; l < HD_MAX_LEVELS; ++l) flags.fast[l] = 0;
  for (int l = 0; l < lv->n_levels; ++l) {
    flags.fast[l] = is_dense_nhwc(*grad_lv, l, B) ? 1 : 0;
    any_fast = any_fast || flags.fast[l];
    const long long total = (ix.first_cell[l + 1] - ix.first_cell[l]) * lv->no;
    max_total = std::max(max_total, flags.fast[l] ? (total + 3) / 4 : total);
  }
  // one workgroup covers 256 work items per sweep; cap at 16 workgroups per CU and grid-stride the rest
  long long gx = (max_total + 255) / 256;
  if (gx > 4096) gx = 4096;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(yolo_grad_dense_kernel, dim3((unsigned)gx, (unsigned)lv->n_levels), dim3(256), 0,
                     (hipStream_t)stream, *lv, *grad_lv, ix, flags, B, *cfg, grad_scale, tobj);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_yolo_loss_bwd_pos(const hd_yolo_levels* lv, const hd_yolo_levels* grad_lv, int B, int M,
                                          const int32_t* slot_cell, const float* slot_tbox, const int32_t* slot_cls,
                                          const hd_yolo_loss_cfg* cfg, const float* grad_scale,
                                          const void* workspace, size_t workspace_bytes, hd_stream stream) {
  hd_status st = bwd_common_checks(lv, grad_lv, B, cfg, grad_scale, workspace, workspace_bytes);
  if (st != HD_OK) return st;
  if (M < 0) return HD_ERR_SHAPE;
  if (M == 0) return HD_OK;
  if (!slot_cell || !slot_tbox || !slot_cls) return HD_ERR_NULL;
  const LevelIndex ix = make_index(*lv, B);
  float* acc = reinterpret_cast<float*>(const_cast<void*>(workspace));
  float* tobj = reinterpret_cast<float*>(reinterpret_cast<char*>(const_cast<void*>(workspace)) + kAccBytes);
  const long long S = (long long)5 * lv->na * B * M;
  dim3 grid((unsigned)((S + 255) / 256), (unsigned)lv->n_levels);
  hipLaunchKernelGGL((yolo_pos_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, *lv, *grad_lv, ix, B, S,
                     slot_cell, reinterpret_cast<const float4*>(slot_tbox), slot_cls, *cfg, grad_scale, acc, tobj);
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_yolo_loss_bwd(const hd_yolo_levels* lv, const hd_yolo_levels* grad_lv, int B, int M,
                                      const int32_t* slot_cell, const float* slot_tbox, const int32_t* slot_cls,
                                      const hd_yolo_loss_cfg* cfg, const float* grad_scale, const void* workspace,
                                      size_t workspace_bytes, hd_stream stream) {
  hd_status st = hd_yolo_loss_bwd_dense(lv, grad_lv, B, cfg, grad_scale, workspace, workspace_bytes, stream);
  if (st != HD_OK) return st;
  return hd_yolo_loss_bwd_pos(lv, grad_lv, B, M, slot_cell, slot_tbox, slot_cls, cfg, grad_scale, workspace,
                              workspace_bytes, stream);
}
