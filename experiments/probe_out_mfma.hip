// probe_out_mfma.hip — go / no-go probe for a matrix-core form of the network's last convolution (F -> 6 channels).
//
// Idea: expand the 9 taps into the GEMM's M side: P[tap*6 + co][pixel] = sum_c W[tap][c][co] * X[pixel][c]  (54 -> 64 rows,
// 84 % of the MFMA work useful instead of 19 % for a padded 32-wide block per tap), then out[y][x][co] = sum_tap P[tap, co]
// at pixel (y + dy - 1, x + dx - 1): 9 shifted adds.  This probe times only the GEMM skeleton: per 32-pixel block 16
// global_load_dwordx4 per lane straight into the B operand's registers (lane = pixel, 64 channels per lane half) and
// 128 v_mfma_f32_32x32x2_f32 with the weights read from LDS; the sums go to a sink.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_out_mfma experiments/probe_out_mfma.hip && /tmp/probe_out_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int F = 128;
constexpr int THREADS = 512;

template <int MODE>
__global__ __launch_bounds__(THREADS, 1) void probe(const float* __restrict__ x, const float* __restrict__ w, float* sink,
                                                    int blocks_total) {
  __shared__ __attribute__((aligned(16))) float w_s[2 * 16 * 2 * 32 * 4];      // [blk][j][kk][m] x 4 channels: 32 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 16 * 2 * 32; i += THREADS)
    *reinterpret_cast<f32x4*>(w_s + 4 * i) = *reinterpret_cast<const f32x4*>(w + 4 * i);
  __syncthreads();
  const int pix = lane & 31, half = lane >> 5;
  const int per_wg = blocks_total / gridDim.x;
  f32x16 acc0, acc1, tot;
  for (int i = 0; i < 16; ++i) tot[i] = 0.f;
  f32x4 xr[2][16];
  auto load_block = [&](int b, f32x4* dst) {
    const float* src = x + (size_t)b * 32 * F + pix * F + half * 64;
#pragma unroll
    for (int j = 0; j < 16; ++j) dst[j] = *reinterpret_cast<const f32x4*>(src + 4 * j);
  };
  const int b0 = blockIdx.x * per_wg + wave;
  load_block(b0, xr[0]);
#pragma unroll 1
  for (int it = 0; it < per_wg / 8; it += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int b = b0 + (it + u) * 8;
      const int bn = (it + u + 1) * 8 < per_wg ? b + 8 : b;
      if (MODE != 3) load_block(bn, xr[u ^ 1]);
      __builtin_amdgcn_sched_barrier(0);
      for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
      // weights: ds_read_b128 two steps ahead of their MFMAs, waits counted by hand
      const unsigned wbase = (unsigned)(size_t)w_s + (half * 32 + pix) * 16;
      f32x4 wa[2], wb[2];
#define W_READ(J, A, B) asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(A), "=&v"(B) : "v"(wbase), "n"((J) * 1024), "n"(16384 + (J) * 1024))
#define W_STEP(J)                                                                                    \
      {                                                                                              \
        if ((J) < 15) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(wa[(J) & 1]), "+v"(wb[(J) & 1]));   \
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wa[(J) & 1]), "+v"(wb[(J) & 1]));            \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                              \
          const float xv = MODE == 3 ? xr[0][J][e] : xr[u][J][e];                                    \
          if (MODE == 2) { acc0[e] += wa[(J) & 1][e] * xv; acc1[e] += wb[(J) & 1][e] * xv; }         \
          else {                                                                                     \
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[(J) & 1][e], xv, acc0, 0, 0, 0);            \
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[(J) & 1][e], xv, acc1, 0, 0, 0); }          \
        }                                                                                            \
        if ((J) + 2 < 16) W_READ((J) + 2 < 16 ? (J) + 2 : 0, wa[(J) & 1], wb[(J) & 1]);              \
      }
      W_READ(0, wa[0], wb[0]);
      W_READ(1, wa[1], wb[1]);
      W_STEP(0) W_STEP(1) W_STEP(2) W_STEP(3) W_STEP(4) W_STEP(5) W_STEP(6) W_STEP(7)
      W_STEP(8) W_STEP(9) W_STEP(10) W_STEP(11) W_STEP(12) W_STEP(13) W_STEP(14) W_STEP(15)
      tot += acc0 + acc1;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += tot[i];
  if (s == 12345.678f) sink[tid] = s;
}

int main() {
  const int n = 512, blocks = n * 32;                      // 512 patches of 32 x 32: 16384 blocks of 32 pixels
  const size_t xe = (size_t)blocks * 32 * F;
  float *x, *w, *sink;
  hipMalloc(&x, xe * 4); hipMalloc(&w, 2 * 16 * 2 * 32 * 4 * 4); hipMalloc(&sink, 4096);
  std::vector<float> h(xe);
  for (size_t i = 0; i < xe; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(x, h.data(), xe * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, h.data(), 2 * 16 * 2 * 32 * 4 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int k = 0; k < 20; ++k) {
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(THREADS), 0, 0, x, w, sink, blocks);
        if (mode == 1) hipLaunchKernelGGL(probe<0>, dim3(512), dim3(THREADS), 0, 0, x, w, sink, blocks);
        if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(THREADS), 0, 0, x, w, sink, blocks);
        if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(THREADS), 0, 0, x, w, sink, blocks);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d (%s): %.1f us per launch\n", mode,
             mode == 0 ? "256 workgroups" : mode == 1 ? "512 workgroups (1 resident per CU)" : mode == 2 ? "loads only (no MFMA)" : "MFMA only (one block loaded)", ms * 50.f);
    }
  }
  printf("err %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
