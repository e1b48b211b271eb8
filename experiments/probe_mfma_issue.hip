// probe_mfma_issue.hip — how fast does a SIMD issue v_mfma_f32_32x32x2_f32 with W waves resident and A independent
// accumulators per wave (operands in registers, no memory)?  Prints cycles of the matrix pipe per MFMA (64 = peak).
//   hipcc --offload-arch=gfx950 -O3 -o build/probe_mfma_issue experiments/probe_mfma_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int LDSR>
__global__ void k(float* sink, int iters, long long* cyc) {
  __shared__ float lds[4096];
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
  float x = threadIdx.x * 0.001f, w = 1.0f + threadIdx.x * 1e-6f;
  lds[threadIdx.x & 4095] = x;
  __syncthreads();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (LDSR) { w += lds[(threadIdx.x + r) & 4095]; }
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, x, acc[a], 0, 0, 0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) s += acc[a][i];
  if (s == 1234.5f) sink[threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, int LDSR>
void run(int threads, const char* name) {
  float* sink; long long* cyc; hipMalloc(&sink, 4096 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, LDSR>), dim3(256), dim3(threads), 0, 0, sink, iters, cyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, LDSR>), dim3(256), dim3(threads), 0, 0, sink, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double mfma_per_simd = (double)iters * 8 * NACC * (threads / 256);
  printf("%-34s waves/SIMD %d acc %d: %.1f ticks per MFMA per SIMD, %.1f TFLOP/s\n", name, threads / 256, NACC, c / mfma_per_simd,
         mfma_per_simd * 1024 * 4096.0 / (ms * 1e-3) / 1e12);
}

int main() {
  run<1, 0>(256, "1 accumulator"); run<2, 0>(256, "2 accumulators"); run<4, 0>(256, "4 accumulators");
  run<1, 0>(512, "1 accumulator"); run<2, 0>(512, "2 accumulators"); run<4, 0>(512, "4 accumulators");
  run<2, 0>(768, "2 accumulators"); run<2, 0>(1024, "2 accumulators");
  run<2, 1>(512, "2 accumulators + LDS read / 2"); run<4, 1>(512, "4 accumulators + LDS read / 4");
  return 0;
}
