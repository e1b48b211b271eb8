// Probe of the LDS-DMA forms used by the staging experiments (run on a gfx950 box):
//   1. buffer_load_dwordx4 ... offen lds : LDS address = M0 + 16*lane ?  What do out-of-range lanes write?
//   2. does the instruction offset move the LDS address, the global address, or both?
//   3. global_load_lds_dwordx4 with a per-lane 64-bit address.
// Build: hipcc --offload-arch=gfx950 -O2 experiments/probe_lds_dma.hip -o build/probe_lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const float* src, int src_bytes, float* dst, int mode) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) smem[i] = -1.f;       // sentinel
  __syncthreads();
  const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(smem + 64);   // byte 256
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
  // lanes 0..47 read piece (63 - lane) (reversed, to show the LDS side is lane-linear); lanes 48..63 are out of range
  unsigned off = lane < 48 ? (63 - lane) * 16 : 0x80000000u;
  if (mode == 0) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds), "v"(off), "s"(rsrc) : "memory");
  } else if (mode == 1) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:32 lds" ::"s"(lds), "v"(off), "s"(rsrc) : "memory");
  } else if (mode == 2) {
    const float* g = src + (63 - lane) * 4;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(g) : "memory");
  } else {
    // exec-masked lanes: only even lanes issue
    if ((lane & 1) == 0)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds), "v"(off), "s"(rsrc) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) dst[i] = smem[i];
}

int main() {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  float *src, *dst;
  hipMalloc(&src, 4096);
  hipMalloc(&dst, 4096);
  hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 4096, 0, src, 1024, dst, mode);
    std::vector<float> o(1024);
    hipMemcpy(o.data(), dst, 4096, hipMemcpyDeviceToHost);
    printf("mode %d: first word of each 16-byte LDS slot (slot = float index / 4), slots 12..84:\n", mode);
    for (int s = 12; s < 84; ++s) printf("%g%c", o[s * 4], (s % 16 == 15) ? '\n' : ' ');
    printf("\n");
  }
  return 0;
}
