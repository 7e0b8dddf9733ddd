// Store-path micro-benchmark (tools/, not product code): which write pattern reaches the fill rate on a 1.64 GB buffer?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_patterns tools/micro/store_patterns.hip && /tmp/store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));

// P1: short-lived waves, block b writes 16 KB contiguous (what a vectorised fill does)
__global__ __launch_bounds__(256) void p1_fill(f4* out, long long n16) {
  long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
  const f4 v = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) if (i + k * 256 < n16) out[i + k * 256] = v;
}
// P2: persistent grid, each WAVE sweeps its own contiguous chunk with 1-KiB stores
__global__ __launch_bounds__(256) void p2_wave_chunks(f4* out, long long n16) {
  const int lane = threadIdx.x & 63;
  const long long W = (long long)gridDim.x * 4, w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long pieces = n16 / 64;  // 1-KiB pieces
  const long long p0 = pieces * w / W, p1 = pieces * (w + 1) / W;
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (long long p = p0; p < p1; ++p) out[p * 64 + lane] = v;
}
// P3: persistent grid, each wave sweeps its own chunk of ROWS of row_f4 float4s (rows need not be multiples of 1 KiB)
__global__ __launch_bounds__(256) void p3_wave_rows(f4* out, long long rows, int row_f4) {
  const int lane = threadIdx.x & 63;
  const long long W = (long long)gridDim.x * 4, w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long r0 = rows * w / W, r1 = rows * (w + 1) / W;
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (long long r = r0; r < r1; ++r)
    for (int c = lane; c < row_f4; c += 64) out[r * row_f4 + c] = v;
}
// P4: one tile of `rpw` rows per wave, as many blocks as tiles (multi-round)
__global__ __launch_bounds__(256) void p4_tiles(f4* out, long long rows, int row_f4, int rpw) {
  const int lane = threadIdx.x & 63;
  const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long r0 = t * rpw, r1 = std::min(rows, r0 + rpw);
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (long long r = r0; r < r1; ++r)
    for (int c = lane; c < row_f4; c += 64) out[r * row_f4 + c] = v;
}
// P5: block-cooperative: a block of 4 waves writes rows r, r+1, r+2, r+3 of a tile of 4*rpw rows together (wave w: rows w, w+4, ..)
__global__ __launch_bounds__(256) void p5_block_rows(f4* out, long long rows, int row_f4, int rpw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.x * 4 * rpw, r1 = std::min(rows, r0 + 4 * rpw);
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (long long r = r0 + wave; r < r1; r += 4)
    for (int c = lane; c < row_f4; c += 64) out[r * row_f4 + c] = v;
}

template <typename F>
float time_ms(F launch, int iters = 10) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  std::vector<float> ts;
  for (int i = 0; i < iters; ++i) {
    hipEventRecord(a);
    launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main() {
  const long long rows = 2ll * 409200;
  for (int M : {500, 512}) {
    const int row_f4 = M / 4;
    const long long n16 = rows * row_f4;
    const double gb = n16 * 16 / 1e9;
    f4* out;
    hipMalloc(&out, n16 * 16);
    printf("M = %d, %.3f GB\n", M, gb);
    float ms = time_ms([&] { p1_fill<<<(unsigned)((n16 + 1023) / 1024), 256>>>(out, n16); });
    printf("  P1 fill-like short waves            %.4f ms %.3f TB/s\n", ms, gb / ms);
    for (int g : {1024, 1536, 2048, 4096}) {
      ms = time_ms([&] { p2_wave_chunks<<<g, 256>>>(out, n16); });
      printf("  P2 persistent wave chunks grid %4d  %.4f ms %.3f TB/s\n", g, ms, gb / ms);
    }
    for (int g : {1024, 1536, 4096}) {
      ms = time_ms([&] { p3_wave_rows<<<g, 256>>>(out, rows, row_f4); });
      printf("  P3 persistent wave rows grid %4d    %.4f ms %.3f TB/s\n", g, ms, gb / ms);
    }
    for (int rpw : {1, 4, 16, 32}) {
      const unsigned g = (unsigned)((rows + 4 * rpw - 1) / (4 * rpw));
      ms = time_ms([&] { p4_tiles<<<g, 256>>>(out, rows, row_f4, rpw); });
      printf("  P4 tiles of %2d rows per wave         %.4f ms %.3f TB/s\n", rpw, ms, gb / ms);
      ms = time_ms([&] { p5_block_rows<<<g, 256>>>(out, rows, row_f4, rpw); });
      printf("  P5 block-interleaved rows, %2d/wave   %.4f ms %.3f TB/s\n", rpw, ms, gb / ms);
    }
    hipFree(out);
  }
  return 0;
}
