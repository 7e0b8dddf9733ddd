// global_load_lds_dwordx4 on gfx950: lane L of a wave writes its 16 bytes to LDS[M0 base + 16 L] (wave-uniform base).
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/micro/lds_direct.hip -o /tmp/lds_direct && /tmp/lds_direct
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void gvoid;
  // each lane fetches the chunk of lane (L ^ 5): a per-lane global address, a fixed LDS slot per lane
  const int from = (blockIdx.x * 256 + wave * 64 + (lane ^ 5)) * 8;
  __builtin_amdgcn_global_load_lds((gvoid*)(src + from), (lds_void*)(smem + wave * 1024), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  *reinterpret_cast<uint4*>(dst + (blockIdx.x * 256 + threadIdx.x) * 8) = *reinterpret_cast<const uint4*>(smem + threadIdx.x * 16);
}
int main() {
  const int n = 256 * 8 * 64;
  std::vector<unsigned short> h(n), o(n);
  for (int i = 0; i < n; ++i) h[i] = (unsigned short)(i * 7 + 3);
  unsigned short *a, *b;
  hipMalloc(&a, n * 2); hipMalloc(&b, n * 2);
  hipMemcpy(a, h.data(), n * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(64), dim3(256), 4096, 0, a, b);
  hipMemcpy(o.data(), b, n * 2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < n / 8; ++t) {
    const int blk = t / 256, tid = t % 256, wave = tid / 64, lane = tid % 64;
    const int from = (blk * 256 + wave * 64 + (lane ^ 5)) * 8;
    for (int e = 0; e < 8; ++e) bad += o[t * 8 + e] != h[from + e];
  }
  printf("lds_direct: %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
  return bad != 0;
}
