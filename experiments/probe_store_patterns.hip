// Store-path probe: what does one 1-KiB wave store instruction cost a CU for the access patterns of the
// convolution epilogues?  256 workgroups x 8 waves, every wave streams stores over its own region.
// Build: hipcc --offload-arch=gfx950 -O2 experiments/probe_store_patterns.hip -o build/probe_store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// pattern: 0 lane-contiguous 16 B (1 KiB contiguous)         1 16-B pieces at 32-B stride, pixel pitch 1 KiB (perm16 fp32)
//          2 64-B runs, pixel pitch 1 KiB (natural-row fp32)   3 64-B runs, pixel pitch 512 B (perm16 bf16)
//          4 8-B pieces -> 32-B runs, pixel pitch 512 B (natural-row bf16, b64)   5 as 1 but both halves back to back
template <int PAT>
__global__ __launch_bounds__(512) void k(char* buf, size_t per_wave, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* base = buf + ((size_t)blockIdx.x * 8 + wave) * per_wave;
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (unsigned)per_wave, 0x00020000);
  const int p = lane & 15, q = lane >> 4;
  u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
  u32x2 v2 = {(unsigned)lane, 1u};
  for (int i = 0; i < iters; ++i) {
    // a new 16-pixel row of a 16 x 1 KiB (or 512 B) strip per iteration
    if (PAT == 0) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, i * 1024 + lane * 16, 0, 0);
    if (PAT == 1) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, i * 16384 + p * 1024 + q * 32, 0, 0);
    if (PAT == 2) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, i * 16384 + p * 1024 + q * 16, 0, 0);
    if (PAT == 3) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, i * 8192 + p * 512 + q * 16, 0, 0);
    if (PAT == 4) __builtin_amdgcn_raw_buffer_store_b64(v2, rsrc, i * 8192 + p * 512 + q * 8, 0, 0);
    if (PAT == 5) {
      __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, i * 16384 + p * 1024 + q * 32, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, i * 16384 + p * 1024 + q * 32 + 16, 0, 0);
    }
  }
}

template <int PAT>
void run(char* buf, size_t per_wave, int iters, const char* name, double bytes_per_iter) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(512), 0, 0, buf, per_wave, iters);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(512), 0, 0, buf, per_wave, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double instr = (double)iters * (PAT == 5 ? 2 : 1) * 8;              // wave instructions per CU
  const double bytes = bytes_per_iter * iters * 8 * 256;
  printf("%-52s %.3f ms  %7.1f GB/s chip  %6.1f ns per wave-instruction and CU  (%.1f B/clk/CU at 2.1 GHz)\n", name, ms,
         bytes / ms / 1e6, ms * 1e6 / instr, bytes / 256 / (ms * 1e-3 * 2.1e9));
}

int main() {
  const int iters = 512;
  const size_t per_wave = (size_t)iters * 16384;          // 8 MiB per wave, 16 GiB total would be too much: reuse below
  char* buf; const size_t total = (size_t)256 * 8 * per_wave;
  if (hipMalloc(&buf, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  run<0>(buf, per_wave, iters, "0 lane-contiguous 1 KiB", 1024);
  run<1>(buf, per_wave, iters, "1 16-B pieces @32 B, pixel pitch 1 KiB (fp32 perm16)", 1024);
  run<2>(buf, per_wave, iters, "2 64-B runs, pixel pitch 1 KiB (fp32 natural)", 1024);
  run<3>(buf, per_wave, iters, "3 64-B runs, pixel pitch 512 B (bf16 perm16)", 1024);
  run<4>(buf, per_wave, iters, "4 32-B runs (b64), pixel pitch 512 B (bf16 natural)", 512);
  run<5>(buf, per_wave, iters, "5 = 1, both halves back to back (full lines in 2)", 2048);
  return 0;
}
