// buffer_soffset_probe.hip — is the SGPR offset (soffset) of a raw buffer instruction part of the hardware range check on gfx950?
// pointwise_bwd_kernel puts a row block's byte offset into soffset and relies on "rows past R read as zeros / their stores are
// dropped".  Everything here stays inside ONE 1 MiB allocation, so neither answer can fault.
//   build: hipcc --offload-arch=gfx950 -O2 -o tools/micro/bin/buffer_soffset_probe tools/micro/buffer_soffset_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void probe(unsigned* base, unsigned* out, int records_bytes, int soff) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, records_bytes, 0x00020000);
  const int lane = threadIdx.x;
  // A: lane offset in range, soffset far past the records
  out[lane] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, soff, 0);
  // B: the same address, all of it in voffset (must read zero)
  out[64 + lane] = __builtin_amdgcn_raw_buffer_load_b32(rs, soff + lane * 4, 0, 0);
  // C: in range either way
  out[128 + lane] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, 0, 0);
  // D: voffset + soffset in range, (must read base[32 + lane % 32])
  out[192 + lane] = __builtin_amdgcn_raw_buffer_load_b32(rs, (lane & 31) * 4, 128, 0);
  // E: lane offset in range but only soffset + voffset crosses the end for the upper half-wave (records 256 B, soffset 128)
  out[256 + lane] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, 128, 0);
  // F: 16-byte load whose last dwords cross the end of the records: voffset 240 + 4 * (lane & 3)
  // stores: value 0xABCD0000 | lane to [soff + lane * 4] through soffset (S1) and through voffset (S2, shifted by 1024)
  __builtin_amdgcn_raw_buffer_store_b32(0xABCD0000u | lane, rs, lane * 4, soff, 0);
  __builtin_amdgcn_raw_buffer_store_b32(0xDCBA0000u | lane, rs, soff + 1024 + lane * 4, 0, 0);
}

int main() {
  const int N = 1 << 18;  // dwords = 1 MiB
  std::vector<unsigned> h(N);
  for (int i = 0; i < N; ++i) h[i] = 0x10000000u + i;
  unsigned *d, *o;
  hipMalloc(&d, N * 4);
  hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
  hipMemset(o, 0xff, 4096);
  const int records = 256, soff = 8192;
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, records, soff);
  hipDeviceSynchronize();
  std::vector<unsigned> r(1024), back(N);
  hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  hipMemcpy(back.data(), d, N * 4, hipMemcpyDeviceToHost);
  auto show = [&](const char* name, int at) { printf("%-58s lane0=%08x lane1=%08x lane33=%08x lane63=%08x\n", name, r[at], r[at + 1], r[at + 33], r[at + 63]); };
  show("A load  voffset in range, soffset past records:", 0);
  show("B load  same address all in voffset (expect 0):", 64);
  show("C load  in range (expect 1000000x):", 128);
  show("D load  voffset+soffset in range:", 192);
  show("E load  records=256 soffset=128 voffset=4*lane:", 256);
  printf("S1 store through soffset past records: mem[soff/4+5] = %08x (was %08x) -> %s\n", back[soff / 4 + 5], 0x10000000u + soff / 4 + 5,
         back[soff / 4 + 5] == (0x10000000u + soff / 4 + 5) ? "DROPPED" : "WRITTEN");
  printf("S2 store through voffset past records: mem[(soff+1024)/4+5] = %08x -> %s\n", back[(soff + 1024) / 4 + 5],
         back[(soff + 1024) / 4 + 5] == (0x10000000u + (soff + 1024) / 4 + 5) ? "DROPPED" : "WRITTEN");
  const bool a_zero = r[0] == 0 && r[1] == 0 && r[63] == 0;
  printf("VERDICT: soffset is %s the range check of raw buffer loads; E upper half %s\n", a_zero ? "INCLUDED in" : "EXCLUDED from",
         r[256 + 33] == 0 ? "zero (sum checked)" : "data (voffset alone checked)");
  return 0;
}
