// pointwise_bwd.hip — the WHOLE backward of a 1x1 convolution on channels-last rows in one launch:
//     dX[R, Cin]    = dY[R, Cout] . W[Cout, Cin]                (the data gradient: reduction over Cout)
//     dW[Cout, Cin] = dY[R, Cout]^T . X[R, Cin]                 (the weight gradient: reduction over the rows)
// from ONE staging of dY (hd_pointwise_conv_dgrad and hd_pointwise_wgrad_partial each read dY, 26 + 31
// launches per YOLOv5s step, and the weight gradient's float32 slab moved as many bytes as its operands).
//
// Shape of the kernel = the streaming data-gradient kernel of pointwise_conv.hip (pointwise_stream_kernel<.., TRB>): a
// workgroup owns one BN-wide column tile of Cin, keeps W[:, tile] (as it lies: [Cout][BN] rows) in LDS for the whole launch,
// and its four waves never meet at a barrier inside the loop — each wave pulls blocks of 32 TM rows of dY coalesced into
// registers (the NEXT block in flight while the current one is multiplied), stages them in a wave-private swizzled LDS area,
// multiplies dX = dY . W through row fragments of that area and transposes the result through a wave-private patch so that
// the stores are 16 bytes per lane along full rows.  New here:
//   * the block's X rows [32 TM, BN] travel with it (registers -> a wave-private padded tile);
//   * dW[:, tile] += dY^T . X from the SAME staged dY rows, read a second time as LDS transpose fragments
//     (ds_read_b64_tr_b16: the reduction index of this product is the row index of both operands).  One swizzle serves both
//     kinds of read: 16 consecutive rows at one 16-byte chunk land on 16 bank groups (row fragments, ds_read_b128), and four
//     consecutive rows at four consecutive chunks land on the four 64-byte quarters of a bank row (transpose fragments);
//   * every wave keeps its [Cout, BN] float32 partial of dW in registers over ALL its row blocks (Cout / 32 x TN
//     accumulator tiles: 128 registers for 256 x 32, 128 x 64 and 64 x 128); at the end the four waves' partials are summed
//     through LDS and ONE partial per workgroup goes to the slab [workgroups per column tile][Cout][Cin] with 16-byte
//     stores.  The slab is (workgroups) x Cout x BN x 4 bytes — 4-8 MB per launch whatever the layer, against
//     slices x Cout x Cin x 4 of the separate weight-gradient kernel (12.8 MB for 6400 x 256 x 256, 33 MB for 25600 x 256 x 256);
//   * ACC: the gradient that reaches x through a shortcut is added to dX in the epilogue (as hd_pointwise_conv_dgrad_add);
//   * workgroups past the main grid sum the row-sliced partials of the PREVIOUS weight-gradient launch (hd_wgrad_job).
// Rows past R are zero in the staged X tile, so they add nothing to dW; their dX rows are never stored.
#include <hip/hip_bf16.h>

#include <type_traits>

#include "hd_bn_fuse.h"
#include "hd_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// [rows][KR] bf16, 16-byte chunks XOR-swizzled by a function of the row that serves row reads AND transpose reads
template <int KR>
struct Stage {
  static constexpr int CPR = KR / 8;
  __device__ static __forceinline__ int f(int row) {
    if (CPR >= 16) return ((row & 3) << 2) | ((row >> 2) & 3);
    if (CPR == 8) return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
    return (row >> 2) & 3;
  }
  __device__ static __forceinline__ int off(int row, int chunk) { return row * (KR * 2) + ((chunk ^ f(row)) << 4); }
};

// transpose fragment (8 consecutive rows m_base + 8 h + j of column col0 + (lane & 31)) of a Stage<KR> image
template <int KR>
__device__ __forceinline__ bf16x8 tr_fragment_sw(const unsigned char* tile, int col0, int m_base, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int row = m_base + 8 * (g >> 1) + q;
  const int col = col0 + 16 * (g & 1) + 4 * p;
  const int chunk = col >> 3, within = (col & 7) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + Stage<KR>::off(row, chunk) + within));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + Stage<KR>::off(row + 4, chunk) + within));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// the same of a plain padded image (row stride `stride` bytes): the W slab and the X tile
__device__ __forceinline__ bf16x8 tr_fragment(const unsigned char* tile, int stride, int col0, int m_base, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const unsigned char* a = tile + (m_base + 8 * (g >> 1) + q) * stride + (col0 + 16 * (g & 1) + 4 * p) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * stride));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int KR, int TN, int TM>
struct BwdLds {
  static constexpr int BN = TN * 32;
  static constexpr int SLAB_ROW = BN * 2 + 64;            // W[:, tile] rows, padded for the transpose reads
  static constexpr int SLAB_BYTES = KR * SLAB_ROW;
  static constexpr int XROW = BN * 2 + (BN == 32 ? 0 : 64);  // 64-byte rows need no padding (four rows = four bank quarters)
  static constexpr int STAGE_BYTES = 32 * TM * KR * 2;
  static constexpr int PATCH_ROW = BN * 2 + 16;
  static constexpr int PATCH_BYTES = 32 * PATCH_ROW;
  static constexpr int AREA0 = STAGE_BYTES > PATCH_BYTES ? STAGE_BYTES : PATCH_BYTES;  // dY stage, then the output patch
  static constexpr int XTILE_BYTES = 32 * TM * XROW;
  static constexpr int WAVE_BYTES = AREA0 + XTILE_BYTES;
  static constexpr int LOOP_BYTES = SLAB_BYTES + 4 * WAVE_BYTES;
  static constexpr int FOLD_BYTES = 4 * KR * BN * 4;       // the four waves' partials of dW[:, tile], summed at the end
  static constexpr int TOTAL = LOOP_BYTES > FOLD_BYTES ? LOOP_BYTES : FOLD_BYTES;
};

template <int KR, int TN, int TM, bool ACC>
__global__ __launch_bounds__(256) void pointwise_bwd_kernel(const unsigned short* __restrict__ dY, long long ld_dy,
                                                            const unsigned short* __restrict__ Wt,   // [KR][N] forward weight
                                                            const unsigned short* __restrict__ X, long long ld_x,
                                                            unsigned short* __restrict__ dX, long long ld_dx, int M, int N,
                                                            const unsigned short* __restrict__ radd, long long ld_r,
                                                            float* __restrict__ slab, int G, const hdfuse::ReduceJob job) {
  using L = BwdLds<KR, TN, TM>;
  constexpr int BN = TN * 32, KS = KR / 16, CPR = KR / 8, NTL = KR / 32;
  constexpr int NLD = TM * KS;           // 16-byte dY loads per lane per row block
  constexpr int XCPR = BN / 8;           // 16-byte chunks per X row
  constexpr int NLX = TM * BN / 16;      // 16-byte X loads per lane per row block (32 TM rows x XCPR chunks over 64 lanes)
  static_assert(64 % CPR == 0 && 64 % XCPR == 0, "a lane's chunk index must not depend on the load index");
  __shared__ float4 s_part[16][17];
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [W slab | 4 wave areas], later the fold area
  if ((int)blockIdx.x >= G) {  // the slice sum of the previous weight-gradient launch rides along (hd_wgrad_job)
    const long long blk = (long long)blockIdx.x - G;
    if (blk < hdfuse::reduce_blocks(job.count)) hdfuse::reduce_block(job, blk, s_part);
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 31, fh = lane >> 5;
  const int NT = N / BN;
  const int lb = (G % 8 == 0) ? ((int)blockIdx.x % 8) * (G / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
  const int nt = lb % NT, n0 = nt * BN;                    // G is a multiple of NT: one column tile per workgroup
  const int slice = lb / NT;                               // this workgroup among those of its column tile
  const int gw = slice * 4 + wave, GW = (G / NT) * 4;
  const int RT = (M + 32 * TM - 1) / (32 * TM);
  const int my = gw < RT ? (RT - 1 - gw) / GW + 1 : 0;

  u32x4 nxt[NLD], nxx[NLX];
  auto load_rows = [&](int rt) {
#pragma unroll
    for (int p = 0; p < NLD; ++p) {
      const int id = p * 64 + lane;
      int m = rt * (32 * TM) + id / CPR;
      m = m < M ? m : M - 1;
      nxt[p] = *reinterpret_cast<const u32x4*>(dY + (long long)m * ld_dy + (id % CPR) * 8);
    }
#pragma unroll
    for (int p = 0; p < NLX; ++p) {
      const int id = p * 64 + lane;
      int m = rt * (32 * TM) + id / XCPR;
      m = m < M ? m : M - 1;
      nxx[p] = *reinterpret_cast<const u32x4*>(X + (long long)m * ld_x + n0 + (id % XCPR) * 8);
    }
  };
  if (my > 0) load_rows(gw);  // in flight while the weight slab is staged
  {
    constexpr int NCH = KR * XCPR, WIT = (NCH + 255) / 256;  // all of a thread's slab loads before its first LDS store
    u32x4 wreg[WIT];
#pragma unroll
    for (int p = 0; p < WIT; ++p) {
      const int id = p * 256 + tid;
      if (NCH % 256 == 0 || id < NCH) wreg[p] = *reinterpret_cast<const u32x4*>(Wt + (long long)(id / XCPR) * N + n0 + (id % XCPR) * 8);
    }
#pragma unroll
    for (int p = 0; p < WIT; ++p) {
      const int id = p * 256 + tid;
      if (NCH % 256 == 0 || id < NCH) *reinterpret_cast<u32x4*>(smem + (id / XCPR) * L::SLAB_ROW + (id % XCPR) * 16) = wreg[p];
    }
  }
  __syncthreads();

  unsigned char* area = smem + L::SLAB_BYTES + wave * L::WAVE_BYTES;  // wave-private: dY stage, then the output patch
  unsigned char* xt = area + L::AREA0;                                 // wave-private X tile
  f32x16 wacc[NTL][TN];
#pragma unroll
  for (int n = 0; n < NTL; ++n)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) wacc[n][j][r] = 0.0f;

  const u32x4 zero = {0u, 0u, 0u, 0u};
  for (int it = 0; it < my; ++it) {
    const int rt = gw + it * GW;
#pragma unroll
    for (int p = 0; p < NLD; ++p) {
      const int id = p * 64 + lane;
      *reinterpret_cast<u32x4*>(area + Stage<KR>::off(id / CPR, id % CPR)) = nxt[p];
    }
#pragma unroll
    for (int p = 0; p < NLX; ++p) {
      const int id = p * 64 + lane, row = id / XCPR;
      *reinterpret_cast<u32x4*>(xt + row * L::XROW + (id % XCPR) * 16) = (rt * (32 * TM) + row < M) ? nxx[p] : zero;
    }
    if (it + 1 < my) load_rows(rt + GW);
    u32x4 racc[ACC ? TM : 1][ACC ? BN / 16 : 1];
    if (ACC) {  // requested now, consumed after the MFMA phase
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < BN / 16; ++q) {
          const int id = q * 64 + lane, row = id / XCPR, c = id % XCPR;
          int m = (rt * TM + i) * 32 + row;
          m = m < M ? m : M - 1;
          racc[i][q] = *reinterpret_cast<const u32x4*>(radd + (long long)m * ld_r + n0 + c * 8);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS writes, before its lanes read each other's
    // ---- dX = dY . W : row fragments of the stage x transpose fragments of the slab
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bf16x8 a[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(area + Stage<KR>::off(i * 32 + frow, 2 * s + fh));
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const bf16x8 b = tr_fragment(smem, L::SLAB_ROW, j * 32, s * 16, lane);
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b, acc[i][j], 0, 0, 0);
      }
    }
    // ---- dW += dY^T . X : transpose fragments of the same stage x transpose fragments of the X tile
#pragma unroll
    for (int ks = 0; ks < 2 * TM; ++ks) {
      bf16x8 bx[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) bx[j] = tr_fragment(xt, L::XROW, j * 32, ks * 16, lane);
#pragma unroll
      for (int n = 0; n < NTL; ++n) {
        const bf16x8 ay = tr_fragment_sw<KR>(area, n * 32, ks * 16, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) wacc[n][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ay, bx[j], wacc[n][j], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // stage reads done before the patch overwrites the area
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m0 = (rt * TM + i) * 32;
      // C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)  ->  patch[row][col] as bf16
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
          const unsigned short v = __builtin_bit_cast(unsigned short, (__bf16)acc[i][j][r]);  // v_cvt_pk_bf16_f32, RNE
          *reinterpret_cast<unsigned short*>(area + row * L::PATCH_ROW + (j * 32 + frow) * 2) = v;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < BN / 16; ++q) {  // 32 rows x BN/8 chunks of 16 bytes over 64 lanes
        const int id = q * 64 + lane, row = id / XCPR, c = id % XCPR;
        uint4 v = *reinterpret_cast<const uint4*>(area + row * L::PATCH_ROW + c * 16);
        if (m0 + row < M) {
          if (ACC) {
            const u32x4 r = racc[i][q];
            const unsigned vv[4] = {v.x, v.y, v.z, v.w}, rr[4] = {r[0], r[1], r[2], r[3]};
            unsigned oo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
              typedef float f32x2 __attribute__((ext_vector_type(2)));
              const f32x2 f = {__uint_as_float(vv[e] << 16) + __uint_as_float(rr[e] << 16),
                               __uint_as_float(vv[e] & 0xffff0000u) + __uint_as_float(rr[e] & 0xffff0000u)};
              oo[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
            }
            v = make_uint4(oo[0], oo[1], oo[2], oo[3]);
          }
          *reinterpret_cast<uint4*>(dX + (long long)(m0 + row) * ld_dx + n0 + c * 8) = v;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // patch reads done before it is overwritten
    }
  }

  // ---- the four waves' partials of dW[:, tile] -> one partial of the workgroup -> slab[slice][KR][N]
  __syncthreads();  // every wave is past its last slab / stage read
  float* fold = reinterpret_cast<float*>(smem) + (size_t)wave * (KR * BN);
#pragma unroll
  for (int n = 0; n < NTL; ++n)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) fold[((n * TN + j) * 16 + r) * 64 + lane] = wacc[n][j][r];
  __syncthreads();
  const float* f0 = reinterpret_cast<const float*>(smem);
  float* out = slab + (size_t)slice * KR * N;
  for (int e = tid; e < KR * BN / 4; e += 256) {
    const int row = e / (BN / 4), c = (e % (BN / 4)) * 4;        // dW row (output channel), first of 4 columns of the tile
    const int n = row >> 5, rr = row & 31;
    const int r = (rr & 3) + 4 * (rr >> 3), h = (rr >> 2) & 1;   // C/D map inverted: row = (r & 3) + 8 (r >> 2) + 4 h
    const int idx = ((n * TN + (c >> 5)) * 16 + r) * 64 + h * 32 + (c & 31);
    const float4 a = *reinterpret_cast<const float4*>(f0 + idx);
    const float4 b = *reinterpret_cast<const float4*>(f0 + KR * BN + idx);
    const float4 c2 = *reinterpret_cast<const float4*>(f0 + 2 * KR * BN + idx);
    const float4 d = *reinterpret_cast<const float4*>(f0 + 3 * KR * BN + idx);
    const float4 s = make_float4((a.x + b.x) + (c2.x + d.x), (a.y + b.y) + (c2.y + d.y), (a.z + b.z) + (c2.z + d.z),
                                 (a.w + b.w) + (c2.w + d.w));
    *reinterpret_cast<float4*>(out + (size_t)row * N + n0 + c) = s;
  }
}

struct BwdPlan {
  int tn, tm, bn, nt, grid, slices;
};

// Column tile and grid of a shape.  Register budget: the wave's dW partial is (Cout / 32) x TN accumulator tiles of 16
// registers, kept at <= 128 registers.  Grid: a multiple of 8 (XCD remap) and of the column-tile count; at most the
// resident cap (one or two workgroups per CU by LDS) and at most one row block per wave.
inline BwdPlan bwd_plan(int64_t R, int Cin, int Cout, int want_grid) {
  BwdPlan p{};
  int tn_max = Cout >= 256 ? 1 : (Cout >= 128 ? 2 : 4);
  int tn = tn_max;
  while (tn > 1 && Cin % (32 * tn)) tn >>= 1;
  p.tn = tn;
  p.tm = Cout <= 64 ? 2 : 1;
  p.bn = 32 * tn;
  p.nt = Cin / p.bn;
  const long long RT = (R + 32 * p.tm - 1) / (32 * p.tm);
  long long L = 8;
  while (L % p.nt) L += 8;
  long long g = (RT + 3) / 4 * p.nt;  // one row block per wave
  g = (g + L - 1) / L * L;
  long long cap = want_grid > 0 ? want_grid : 256;
  if (cap < L) cap = L;
  if (g > cap) g = cap / L * L;
  p.grid = (int)g;
  p.slices = p.grid / p.nt;
  return p;
}

inline bool bwd_shape_ok(int64_t R, int Cin, int Cout) {
  if (!(R >= 1 && R <= 0x7fffffff)) return false;
  if (!(Cout == 32 || Cout == 64 || Cout == 128 || Cout == 256)) return false;
  if (Cin < 32 || Cin % 32 || Cin > 8192) return false;
  const BwdPlan p = bwd_plan(R, Cin, Cout, 0);
  return p.nt <= 32;
}

int g_bwd_grid_override = 0;  // hd_pointwise_bwd_tune: experiments only (0 = the plan's own choice)

}  // namespace

extern "C" int hd_pointwise_bwd_supported(int64_t R, int Cin, int Cout) { return bwd_shape_ok(R, Cin, Cout) ? 1 : 0; }

extern "C" void hd_pointwise_bwd_tune(int grid) { g_bwd_grid_override = grid; }

extern "C" size_t hd_pointwise_bwd_workspace_bytes(int64_t R, int Cin, int Cout) {
  if (!bwd_shape_ok(R, Cin, Cout)) return 0;
  const BwdPlan p = bwd_plan(R, Cin, Cout, g_bwd_grid_override);
  return (size_t)p.slices * Cin * Cout * sizeof(float);
}

extern "C" hd_status hd_pointwise_bwd(const void* dy, int64_t dy_row_stride, const void* w, const void* x, int64_t x_row_stride,
                                      int64_t R, int Cin, int Cout, const void* add, int64_t add_row_stride, void* dx,
                                      int64_t dx_row_stride, void* dw, void* workspace, size_t workspace_bytes,
                                      const hd_wgrad_job* piggyback, hd_wgrad_job* job_out, hd_stream stream) {
  if (!dy || !w || !x || !dx || !dw || !workspace || !job_out) return HD_ERR_NULL;
  if (!bwd_shape_ok(R, Cin, Cout)) return HD_ERR_SHAPE;
  if (dy_row_stride == 0) dy_row_stride = Cout;
  if (x_row_stride == 0) x_row_stride = Cin;
  if (dx_row_stride == 0) dx_row_stride = Cin;
  if (add && add_row_stride == 0) add_row_stride = Cin;
  if (dy_row_stride < Cout || x_row_stride < Cin || dx_row_stride < Cin || dy_row_stride % 8 || x_row_stride % 8 || dx_row_stride % 8)
    return HD_ERR_SHAPE;
  if (add && (add_row_stride < Cin || add_row_stride % 8)) return HD_ERR_SHAPE;
  if (!hd_aligned16(dy) || !hd_aligned16(w) || !hd_aligned16(x) || !hd_aligned16(dx) || !hd_aligned16(dw) || !hd_aligned16(workspace) ||
      (add && !hd_aligned16(add)))
    return HD_ERR_ALIGN;
  const BwdPlan p = bwd_plan(R, Cin, Cout, g_bwd_grid_override);
  if (workspace_bytes < (size_t)p.slices * Cin * Cout * sizeof(float)) return HD_ERR_WORKSPACE;
  hdfuse::ReduceJob jv{};
  if (piggyback) {
    if (!piggyback->slab || !piggyback->dw) return HD_ERR_NULL;
    if (piggyback->slices < 1 || piggyback->count < 4 || piggyback->count % 4) return HD_ERR_SHAPE;
    if (!hd_aligned16(piggyback->slab) || (reinterpret_cast<uintptr_t>(piggyback->dw) & 7u)) return HD_ERR_ALIGN;
    jv.slab = piggyback->slab;
    jv.slices = piggyback->slices;
    jv.count = piggyback->count;
    jv.dw = static_cast<unsigned short*>(piggyback->dw);
  }
  const unsigned extra = jv.slab ? (unsigned)hdfuse::reduce_blocks(jv.count) : 0u;
  hipStream_t s = (hipStream_t)stream;
  const auto* a = static_cast<const unsigned short*>(dy);
  const auto* b = static_cast<const unsigned short*>(w);
  const auto* c = static_cast<const unsigned short*>(x);
  const auto* r = static_cast<const unsigned short*>(add);
  auto* o = static_cast<unsigned short*>(dx);
  float* slab = static_cast<float*>(workspace);
  const int M = (int)R;
  const dim3 grid((unsigned)p.grid + extra);
#define HD_BWD_LAUNCH(KRR, TNN, TMM)                                                                                          \
  do {                                                                                                                        \
    const size_t lds = BwdLds<KRR, TNN, TMM>::TOTAL;                                                                          \
    if (add)                                                                                                                  \
      (void)hd_launch(pointwise_bwd_kernel<KRR, TNN, TMM, true>, grid, dim3(256), lds, s, a, (long long)dy_row_stride, b, c,  \
                      (long long)x_row_stride, o, (long long)dx_row_stride, M, Cin, r, (long long)add_row_stride, slab, p.grid, jv); \
    else                                                                                                                      \
      (void)hd_launch(pointwise_bwd_kernel<KRR, TNN, TMM, false>, grid, dim3(256), lds, s, a, (long long)dy_row_stride, b, c, \
                      (long long)x_row_stride, o, (long long)dx_row_stride, M, Cin, r, 0ll, slab, p.grid, jv);                \
  } while (0)
  if (Cout == 256) HD_BWD_LAUNCH(256, 1, 1);
  else if (Cout == 128) { if (p.tn == 2) HD_BWD_LAUNCH(128, 2, 1); else HD_BWD_LAUNCH(128, 1, 1); }
  else if (Cout == 64) { if (p.tn == 4) HD_BWD_LAUNCH(64, 4, 2); else if (p.tn == 2) HD_BWD_LAUNCH(64, 2, 2); else HD_BWD_LAUNCH(64, 1, 2); }
  else { if (p.tn == 4) HD_BWD_LAUNCH(32, 4, 2); else if (p.tn == 2) HD_BWD_LAUNCH(32, 2, 2); else HD_BWD_LAUNCH(32, 1, 2); }
#undef HD_BWD_LAUNCH
  HD_RETURN_IF_LAUNCH_FAILED();
  job_out->slab = slab;
  job_out->slices = p.slices;
  job_out->count = (int64_t)Cin * Cout;
  job_out->dw = dw;
  return HD_OK;
}
