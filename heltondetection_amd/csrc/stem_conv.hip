// stem_conv.hip — the YOLOv5 v6.x stem: Conv2d(3, Cout, k = 6, s = 2, p = 2) on the image batch, forward (with the
// BatchNorm statistics of the layer that follows in the epilogue) and weight gradient.  The input needs no gradient.
//
// Why hand-written.  With 3 input channels the library's implicit-GEMM kernels run this layer at 105 us forward and
// 139 us weight gradient (+ a float32 -> bf16 cast of the images and the split-K zero-fill / cast launches) on YOLOv5s
// 640^2 bs16: 0.27 ms of a 6.2 ms step for 11 GFLOP that the matrix cores do in 5 us.  The layer is memory-bound — 79 MB of
// float32 pixels in, 105 MB of bf16 activations out — so the kernels below are built around reading every image row once:
//   * a workgroup walks a run of consecutive output rows of one image with a ring of input rows in LDS (output row oy reads
//     input rows 2 oy - 2 .. 2 oy + 3; advancing one output row brings in two new rows, loaded as float32 or bf16 straight
//     from the image tensor, converted to bf16 on the way into LDS — no separate cast pass);
//   * channels-last pixels make the six taps of one kernel row 18 CONSECUTIVE bf16 (6 px x 3 ch) of the LDS row, so the
//     implicit GEMM's reduction index is laid out as 6 kernel rows x 24 (18 real + 6 whose weights are zero): 144 = nine
//     k-steps of v_mfma_f32_32x32x16_bf16, and an operand fragment (8 consecutive k) is 16 contiguous bytes at
//     12 ox + const — no im2col, no gather; the 12-byte pixel-pair stride walks the 32 lanes over all banks;
//   * the weights ([Cout][6][6][3] -> [Cout][144], 9 x Cout/32 fragments) live in registers for the whole kernel.
// Weight gradient: dW[n][k] = sum over pixels dY[p][n] * patch[p][k] with the SAME ring; dY rows are staged [px][Cout] and
// read through LDS transpose reads (the reduction index is the pixel), the patch operand is eight 2-byte reads 12 bytes apart
// per fragment.  Every wave keeps its own [Cout x 160] accumulators over the workgroup's rows; partials go to a float32
// slab [workgroups][Cout][108] and the usual slice sum (hd_wgrad_job) finishes the gradient.
#include <hip/hip_bf16.h>

#include "hd_bn_fuse.h"
#include "hd_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int kStatReplicas = 16;  // == hd_bn_act_replicas()
constexpr int kRing = 8;           // input rows resident per workgroup: 6 in use + 2 arriving
constexpr int kPadPx = 4;          // zero pixels left of the image in an LDS row (>= 2; 4 keeps 8-byte alignment of the stores)
constexpr int KP = 144;            // padded reduction length: 6 kernel rows x 24

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 f = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}

// left pad kPadPx pixels, right pad 4 pixels + slack for the output pixels past the row end that a 32-pixel tile still
// touches (the weight gradient reads them unclamped: their dY rows are zero, the slack is zero-filled and never written)
__host__ __device__ inline int ring_row_bytes(int W) { return ((W + kPadPx + 4 + 64) * 6 + 15) / 16 * 16; }

// k (0..143) -> index into the real [6][6][3] kernel (0..107), or -1 for the zero-weight padding slots
__device__ __forceinline__ int real_k(int k) {
  const int ky = k / 24, e = k - ky * 24;
  return e < 18 ? ky * 18 + e : -1;
}

template <typename TIN>
struct RowLoader;
// float32 image rows: W * 3 floats = W * 3 / 4 float4 (W % 4 == 0)
template <>
struct RowLoader<float> {
  static constexpr int PER = 4;  // elements per load
  typedef float4 Vec;
  static __device__ __forceinline__ void stash(unsigned char* dst, const Vec v) {
    *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
  }
  static constexpr int BYTES = 8;  // LDS bytes per load
};
template <>
struct RowLoader<unsigned short> {  // bf16 image rows
  static constexpr int PER = 4;
  typedef uint2 Vec;
  static __device__ __forceinline__ void stash(unsigned char* dst, const Vec v) { *reinterpret_cast<uint2*>(dst) = v; }
  static constexpr int BYTES = 8;
};

// Shared ring logic: image row `iy` of image b (zeros outside the image) -> ring slot.  A row is W * 3 elements = NV vectors
// of 4; thread t takes vectors t, t + 256, ... (<= MAXV per thread).
template <typename TIN, int MAXV>
struct Ring {
  typedef typename RowLoader<TIN>::Vec Vec;
  unsigned char* base;
  int row_bytes, W, H, nv;
  const TIN* x;  // image b
  __device__ __forceinline__ unsigned char* slot(int iy) const { return base + ((iy + 2) & (kRing - 1)) * row_bytes; }
  __device__ __forceinline__ void load(int iy, Vec* r) const {
    const int ic = min(max(iy, 0), H - 1);
    const Vec* src = reinterpret_cast<const Vec*>(x + (long long)ic * W * 3);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int id = i * 256 + threadIdx.x;
      if (id < nv) r[i] = src[id];
    }
  }
  __device__ __forceinline__ void stash(int iy, const Vec* r) const {
    unsigned char* dst = slot(iy) + kPadPx * 6;
    const bool in = iy >= 0 && iy < H;
    Vec zero;
    __builtin_memset(&zero, 0, sizeof(Vec));
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int id = i * 256 + threadIdx.x;
      if (id < nv) RowLoader<TIN>::stash(dst + id * RowLoader<TIN>::BYTES, in ? r[i] : zero);
    }
  }
};

// operand fragment of output pixel ox for k-step ks (0..8): 8 consecutive bf16 of one kernel row, 4-byte aligned
__device__ __forceinline__ bf16x8 patch_fragment(const unsigned char* ring_base, int row_bytes, int oy, int ox, int ks, int h) {
  const int e48 = (ks % 3) * 16 + 8 * h;                 // position inside the pair of kernel rows 2q, 2q + 1
  const int ky = 2 * (ks / 3) + (e48 >= 24 ? 1 : 0), off = e48 >= 24 ? e48 - 24 : e48;
  const int iy = 2 * oy - 2 + ky;
  const unsigned char* p = ring_base + ((iy + 2) & (kRing - 1)) * row_bytes + (2 * ox - 2 + kPadPx) * 6 + off * 2;
  const unsigned* q = reinterpret_cast<const unsigned*>(p);
  const u32x4 v = {q[0], q[1], q[2], q[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <typename TIN, int NT, bool STATS>
__global__ __launch_bounds__(256) void stem6x6_fwd_kernel(const TIN* __restrict__ X, const unsigned short* __restrict__ Wt,
                                                          unsigned short* __restrict__ Y, long long ld_y, int B, int H, int W,
                                                          int rows_per_wg, float* __restrict__ stats) {
  constexpr int COUT = NT * 32;
  constexpr int PATCH_ROW = COUT * 2 + 16, PATCH = 32 * PATCH_ROW;
  constexpr int MAXV = 4;  // W * 3 / 4 <= 1024 vectors: W <= 1364
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [ring | weights [COUT][144] | 4 patches]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 31, fh = lane >> 5;
  const int Ho = H / 2, Wo = W / 2;
  const int row_bytes = ring_row_bytes(W);
  unsigned char* ws = smem + kRing * row_bytes;
  unsigned char* patch = ws + COUT * KP * 2 + wave * PATCH;
  const int spi = (Ho + rows_per_wg - 1) / rows_per_wg;
  const int b = blockIdx.x / spi;
  const int oy0 = (blockIdx.x % spi) * rows_per_wg, oy1 = min(Ho, oy0 + rows_per_wg);

  for (int i = tid; i < kRing * row_bytes / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < COUT * KP; i += 256) {
    const int n = i / KP, rk = real_k(i % KP);
    reinterpret_cast<unsigned short*>(ws)[i] = rk >= 0 ? Wt[n * 108 + rk] : (unsigned short)0;
  }
  __syncthreads();
  bf16x8 wf[9][NT];
#pragma unroll
  for (int ks = 0; ks < 9; ++ks)
#pragma unroll
    for (int j = 0; j < NT; ++j) wf[ks][j] = *reinterpret_cast<const bf16x8*>(ws + ((j * 32 + frow) * KP + ks * 16 + 8 * fh) * 2);

  Ring<TIN, MAXV> ring;
  ring.base = smem;
  ring.row_bytes = row_bytes;
  ring.W = W;
  ring.H = H;
  ring.nv = W * 3 / 4;
  ring.x = X + (long long)b * H * W * 3;
  typename RowLoader<TIN>::Vec ra[MAXV], rb[MAXV];
  // rows 2 oy0 - 2 .. 2 oy0 + 3 for the first output row, then two rows ahead
  {  // rows 2 oy0 - 2 .. 2 oy0 + 3, three loads in flight at a time (six serial round trips would cost ~5 us per workgroup)
    typename RowLoader<TIN>::Vec rc[MAXV];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int iy = 2 * oy0 - 2 + 3 * half;
      ring.load(iy, ra);
      ring.load(iy + 1, rb);
      ring.load(iy + 2, rc);
      ring.stash(iy, ra);
      ring.stash(iy + 1, rb);
      ring.stash(iy + 2, rc);
    }
  }
  ring.load(2 * oy0 + 4, ra);
  ring.load(2 * oy0 + 5, rb);

  float s_sum[NT], s_sq[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) s_sum[j] = s_sq[j] = 0.0f;
  const int tiles = (Wo + 31) / 32;
  for (int oy = oy0; oy < oy1; ++oy) {
    __syncthreads();  // rows 2 oy - 2 .. 2 oy + 3 are visible; every wave is past output row oy - 1
    if (oy + 1 < oy1) {
      // rows 2 oy + 4, 2 oy + 5 (for the NEXT output row) go into the slots rows 2 oy - 4, 2 oy - 3 left long ago — no slot of
      // the six live rows — and the loads of the two rows after them travel while this row is multiplied
      ring.stash(2 * oy + 4, ra);
      ring.stash(2 * oy + 5, rb);
      ring.load(2 * oy + 6, ra);
      ring.load(2 * oy + 7, rb);
    }
    for (int t = wave; t < tiles; t += 4) {
      const int ox = min(t * 32 + frow, Wo - 1);  // lanes past the row end recompute its last pixel; never stored
      f32x16 acc[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
#pragma unroll
      for (int ks = 0; ks < 9; ++ks) {
        const bf16x8 a = patch_fragment(smem, row_bytes, oy, ox, ks, fh);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wf[ks][j], acc[j], 0, 0, 0);
      }
      // C/D map: col (channel) = lane & 31, row (pixel) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)  ->  patch[pixel][channel]
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int prow = (r & 3) + 8 * (r >> 2) + 4 * fh;
          const unsigned short v = __builtin_bit_cast(unsigned short, (__bf16)acc[j][r]);
          *reinterpret_cast<unsigned short*>(patch + prow * PATCH_ROW + (j * 32 + frow) * 2) = v;
          if (STATS && t * 32 + prow < Wo) {
            const float f = __uint_as_float((unsigned)v << 16);
            s_sum[j] += f;
            s_sq[j] = fmaf(f, f, s_sq[j]);
          }
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < NT * 2; ++q) {  // 32 pixels x COUT / 8 chunks of 16 bytes over 64 lanes
        const int id = q * 64 + lane, prow = id / (COUT / 8), ch = id % (COUT / 8);
        const uint4 v = *reinterpret_cast<const uint4*>(patch + prow * PATCH_ROW + ch * 16);
        const int px = t * 32 + prow;
        if (px < Wo) *reinterpret_cast<uint4*>(Y + ((long long)(b * Ho + oy) * Wo + px) * ld_y + ch * 8) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }

  if (STATS) {
    __syncthreads();
    float* s_st = reinterpret_cast<float*>(smem);  // [2][COUT]
    for (int i = tid; i < 2 * COUT; i += 256) s_st[i] = 0.0f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float a = s_sum[j] + __shfl_xor(s_sum[j], 32);
      const float q = s_sq[j] + __shfl_xor(s_sq[j], 32);
      if (fh == 0) {
        atomicAdd(&s_st[j * 32 + frow], a);
        atomicAdd(&s_st[COUT + j * 32 + frow], q);
      }
    }
    __syncthreads();
    float* dst = stats + (size_t)(blockIdx.x % kStatReplicas) * 2 * COUT;
    for (int i = tid; i < 2 * COUT; i += 256) atomicAdd(&dst[i], s_st[i]);
  }
}

// LDS transpose read of an A-type fragment from a [reduction row][column] bf16 tile (csrc/pointwise_conv.hip: tr_fragment)
__device__ __forceinline__ bf16x8 tr_fragment(const unsigned char* tile, int stride, int col0, int m_base, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const unsigned char* a = tile + (m_base + 8 * (g >> 1) + q) * stride + (col0 + 16 * (g & 1) + 4 * p) * 2;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 4 * stride));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// dY row (Wo pixels x CY 16-byte chunks) -> registers, all loads issued back to back; then -> LDS [pixel][COUT]
template <int MAXY, int CY>
__device__ __forceinline__ void y_row_load(const unsigned short* __restrict__ src, long long ld, int ny, u32x4* r) {
#pragma unroll
  for (int i = 0; i < MAXY; ++i) {
    const int id = min(i * 256 + (int)threadIdx.x, ny - 1);  // threads past the row end re-read its last chunk (never stashed)
    r[i] = *reinterpret_cast<const u32x4*>(src + (long long)(id / CY) * ld + (id % CY) * 8);
  }
}
template <int MAXY, int CY>
__device__ __forceinline__ void y_row_stash(unsigned char* ys, int SY, int ny, const u32x4* r) {
#pragma unroll
  for (int i = 0; i < MAXY; ++i) {
    const int id = i * 256 + threadIdx.x;
    if (id < ny) *reinterpret_cast<u32x4*>(ys + (id / CY) * SY + (id % CY) * 16) = r[i];
  }
}

// dW partials.  MFMA roles: D[n][kk] += sum_p A[n][p] B[p][kk], A = dY^T (transpose read of the staged dY row), B = the
// patch operand: lane (kk = lane & 31, pixels 8 h .. 8 h + 7) reads one 2-byte element per pixel, 12 bytes apart.
template <typename TIN, int NT, int MAXY>
__global__ __launch_bounds__(256, (NT == 1 ? 2 : 1)) void stem6x6_wgrad_kernel(const TIN* __restrict__ X, const unsigned short* __restrict__ dY,
                                                            long long ld_dy, int B, int H, int W, int rows_per_wg,
                                                            float* __restrict__ slab) {
  constexpr int COUT = NT * 32;
  constexpr int MAXV = 4;
  constexpr int KT = 5;                    // k tiles of 32: 144 -> 160
  // dY row stride in LDS (bytes): the four 64-byte rows a half-wave's transpose read touches must fall on disjoint bank
  // quarters, i.e. stride = 64 or 192 (mod 256): dense for 32 channels, 64 bytes of padding for 64
  constexpr int SY = COUT == 32 ? 64 : COUT * 2 + 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [ring | dY row: Wo16 pixels x SY]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 31, fh = lane >> 5;
  const int Ho = H / 2, Wo = W / 2, Wo16 = (Wo + 31) / 32 * 32;
  const int row_bytes = ring_row_bytes(W);
  unsigned char* ys = smem + kRing * row_bytes;
  const int spi = (Ho + rows_per_wg - 1) / rows_per_wg;
  const int b = blockIdx.x / spi;
  const int oy0 = (blockIdx.x % spi) * rows_per_wg, oy1 = min(Ho, oy0 + rows_per_wg);

  for (int i = tid; i < (kRing * row_bytes + Wo16 * SY) / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();

  Ring<TIN, MAXV> ring;
  ring.base = smem;
  ring.row_bytes = row_bytes;
  ring.W = W;
  ring.H = H;
  ring.nv = W * 3 / 4;
  ring.x = X + (long long)b * H * W * 3;
  typename RowLoader<TIN>::Vec ra[MAXV], rb[MAXV];
  constexpr int CY = COUT / 8;  // 16-byte chunks per dY pixel
  // MAXY: dY chunks per thread per row, 6 (Wo * CY <= 1536) or 12 (<= 3072, hd_stem_conv_supported)
  const int ny = Wo * CY;
  u32x4 ry[MAXY];
  const unsigned short* dy_img = dY + (long long)b * Ho * Wo * ld_dy;
#define load_y(OY) y_row_load<MAXY, CY>(dy_img + (long long)min((OY), Ho - 1) * Wo * ld_dy, ld_dy, ny, ry)
#define stash_y() y_row_stash<MAXY, CY>(ys, SY, ny, ry)
  {
    typename RowLoader<TIN>::Vec rc[MAXV];
    load_y(oy0);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int iy = 2 * oy0 - 2 + 3 * half;
      ring.load(iy, ra);
      ring.load(iy + 1, rb);
      ring.load(iy + 2, rc);
      ring.stash(iy, ra);
      ring.stash(iy + 1, rb);
      ring.stash(iy + 2, rc);
    }
  }
  stash_y();
  ring.load(2 * oy0 + 4, ra);
  ring.load(2 * oy0 + 5, rb);
  load_y(oy0 + 1);

  f32x16 acc[NT][KT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.0f;

  // this lane's patch column per k tile: kk = 32 t + frow -> (kernel row, byte offset inside the row's 18 elements) + the
  // lane half's 8-pixel offset; padding slots read neighbouring pixels' data: their dW entries are never written out
  int ky_[KT], ob_[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const int k = min(32 * t + frow, KP - 1);
    ky_[t] = k / 24;
    ob_[t] = (k % 24) * 2 + (kPadPx - 2) * 6 + 8 * fh * 12;
  }
  const int steps = Wo16 / 16;  // 16 pixels per k-step
  for (int oy = oy0; oy < oy1; ++oy) {
    __syncthreads();
    const unsigned char* rowp_[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) rowp_[t] = smem + ((2 * oy - 2 + ky_[t] + 2) & (kRing - 1)) * row_bytes + ob_[t];
    for (int st = wave; st < steps; st += 4) {
      const int px0 = st * 16;
      bf16x8 a[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) a[j] = tr_fragment(ys, SY, j * 32, px0, lane);
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        // eight 2-byte reads 12 bytes apart from one base address (immediate offsets); pixels past the row end fall into
        // the zero slack of the ring row, and their dY rows are zero anyway
        const unsigned char* base = rowp_[t] + px0 * 12;
        u32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned lo = *reinterpret_cast<const unsigned short*>(base + (2 * q) * 12);
          const unsigned hi = *reinterpret_cast<const unsigned short*>(base + (2 * q + 1) * 12);
          v[q] = lo | (hi << 16);
        }
        const bf16x8 bq = __builtin_bit_cast(bf16x8, v);
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], bq, acc[j][t], 0, 0, 0);
      }
    }
    __syncthreads();  // the dY row has been read by every wave
    if (oy + 1 < oy1) {
      ring.stash(2 * oy + 4, ra);
      ring.stash(2 * oy + 5, rb);
      stash_y();
      ring.load(2 * oy + 6, ra);
      ring.load(2 * oy + 7, rb);
      load_y(oy + 2);
    }
  }

#undef load_y
#undef stash_y
  // the four waves' accumulators -> one partial per workgroup (LDS float atomics into [COUT][160]), then the real 108
  __syncthreads();
  float* s_acc = reinterpret_cast<float*>(smem);  // COUT * 160 floats <= 40 KB: inside the ring + dY area for every supported W
  for (int i = tid; i < COUT * KT * 32; i += 256) s_acc[i] = 0.0f;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = j * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;  // C/D map: row = n, col = kk
        atomicAdd(&s_acc[n * (KT * 32) + t * 32 + frow], acc[j][t][r]);
      }
  __syncthreads();
  float* out = slab + (size_t)blockIdx.x * COUT * 108;
  for (int i = tid; i < COUT * KP; i += 256) {
    const int n = i / KP, rk = real_k(i % KP);
    if (rk >= 0) out[n * 108 + rk] = s_acc[n * (KT * 32) + (i % KP)];
  }
}

struct StemPlan {
  int rows_per_wg, wgs;
};
inline StemPlan stem_plan(int B, int Ho) {
  // ~2 workgroups per CU, slices never cross an image; at least 4 output rows per workgroup (6 ring rows of start-up)
  StemPlan p;
  int spi = 512 / B;
  if (spi < 1) spi = 1;
  int rows = (Ho + spi - 1) / spi;
  if (rows < 4) rows = 4;
  p.rows_per_wg = rows;
  p.wgs = B * ((Ho + rows - 1) / rows);
  return p;
}

}  // namespace

extern "C" int hd_stem_conv_supported(int B, int H, int W, int Cout, int x_dtype) {
  return B >= 1 && H >= 8 && W >= 8 && H % 2 == 0 && W % 4 == 0 && W <= 1344 && (Cout == 32 || Cout == 64) &&
         (W / 2) * (Cout / 8) <= 3072 /* dY row prefetch of the weight gradient */ && (x_dtype == 0 || x_dtype == 1) &&
         (long long)B * H * W <= 0x7fffffff;
}

extern "C" size_t hd_stem_conv_wgrad_workspace_bytes(int B, int H, int W, int Cout) {
  if (!hd_stem_conv_supported(B, H, W, Cout, 0)) return 0;
  return (size_t)stem_plan(B, H / 2).wgs * Cout * 108 * sizeof(float);
}

extern "C" hd_status hd_stem_conv_fwd(const void* x, int x_dtype, const void* w, int B, int H, int W, int Cout, void* y,
                                      int64_t y_row_stride, float* stats, hd_stream stream) {
  if (!x || !w || !y) return HD_ERR_NULL;
  if (!hd_stem_conv_supported(B, H, W, Cout, x_dtype)) return HD_ERR_SHAPE;
  if (y_row_stride == 0) y_row_stride = Cout;
  if (y_row_stride < Cout || y_row_stride % 8) return HD_ERR_SHAPE;
  if (!hd_aligned16(x) || !hd_aligned16(y) || (reinterpret_cast<uintptr_t>(w) & 1u)) return HD_ERR_ALIGN;
  if (stats && hd_bn_act_replicas() != kStatReplicas) return HD_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const StemPlan p = stem_plan(B, H / 2);
  const size_t lds = (size_t)kRing * ring_row_bytes(W) + (size_t)Cout * KP * 2 + 4u * 32u * (Cout * 2 + 16);
  const auto* wt = static_cast<const unsigned short*>(w);
  auto* yo = static_cast<unsigned short*>(y);
#define HD_STEM_FWD(TIN, NTT)                                                                                                  \
  do {                                                                                                                         \
    if (stats) (void)hd_launch(stem6x6_fwd_kernel<TIN, NTT, true>, dim3((unsigned)p.wgs), dim3(256), lds, s,                   \
                               static_cast<const TIN*>(x), wt, yo, y_row_stride, B, H, W, p.rows_per_wg, stats);               \
    else (void)hd_launch(stem6x6_fwd_kernel<TIN, NTT, false>, dim3((unsigned)p.wgs), dim3(256), lds, s,                        \
                         static_cast<const TIN*>(x), wt, yo, y_row_stride, B, H, W, p.rows_per_wg, (float*)nullptr);           \
  } while (0)
  if (x_dtype == 0) {
    if (Cout == 32) HD_STEM_FWD(float, 1);
    else HD_STEM_FWD(float, 2);
  } else {
    if (Cout == 32) HD_STEM_FWD(unsigned short, 1);
    else HD_STEM_FWD(unsigned short, 2);
  }
#undef HD_STEM_FWD
  HD_RETURN_IF_LAUNCH_FAILED();
  return HD_OK;
}

extern "C" hd_status hd_stem_conv_wgrad_partial(const void* dy, int64_t dy_row_stride, const void* x, int x_dtype, int B, int H,
                                                int W, int Cout, void* dw, void* workspace, size_t workspace_bytes,
                                                hd_wgrad_job* job_out, hd_stream stream) {
  if (!dy || !x || !dw || !workspace || !job_out) return HD_ERR_NULL;
  if (!hd_stem_conv_supported(B, H, W, Cout, x_dtype)) return HD_ERR_SHAPE;
  if (dy_row_stride == 0) dy_row_stride = Cout;
  if (dy_row_stride < Cout || dy_row_stride % 8) return HD_ERR_SHAPE;
  if (!hd_aligned16(dy) || !hd_aligned16(x) || !hd_aligned16(workspace) || (reinterpret_cast<uintptr_t>(dw) & 7u)) return HD_ERR_ALIGN;
  const StemPlan p = stem_plan(B, H / 2);
  if (workspace_bytes < (size_t)p.wgs * Cout * 108 * sizeof(float)) return HD_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int Wo16 = (W / 2 + 31) / 32 * 32;
  size_t lds = (size_t)kRing * ring_row_bytes(W) + (size_t)Wo16 * (Cout == 32 ? 64 : Cout * 2 + 64);
  if (lds < (size_t)Cout * 160 * sizeof(float)) lds = (size_t)Cout * 160 * sizeof(float);  // the fold area
  const auto* g = static_cast<const unsigned short*>(dy);
  float* slab = static_cast<float*>(workspace);
#define HD_STEM_WG(TIN, NTT)                                                                                                 \
  do {                                                                                                                       \
    if ((W / 2) * (Cout / 8) <= 1536)                                                                                        \
      (void)hd_launch(stem6x6_wgrad_kernel<TIN, NTT, 6>, dim3((unsigned)p.wgs), dim3(256), lds, s, static_cast<const TIN*>(x), \
                      g, dy_row_stride, B, H, W, p.rows_per_wg, slab);                                                       \
    else                                                                                                                     \
      (void)hd_launch(stem6x6_wgrad_kernel<TIN, NTT, 12>, dim3((unsigned)p.wgs), dim3(256), lds, s,                          \
                      static_cast<const TIN*>(x), g, dy_row_stride, B, H, W, p.rows_per_wg, slab);                           \
  } while (0)
  if (x_dtype == 0) {
    if (Cout == 32) HD_STEM_WG(float, 1);
    else HD_STEM_WG(float, 2);
  } else {
    if (Cout == 32) HD_STEM_WG(unsigned short, 1);
    else HD_STEM_WG(unsigned short, 2);
  }
#undef HD_STEM_WG
  HD_RETURN_IF_LAUNCH_FAILED();
  job_out->slab = slab;
  job_out->slices = p.wgs;
  job_out->count = (int64_t)Cout * 108;
  job_out->dw = dw;
  return HD_OK;
}
