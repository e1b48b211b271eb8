/* c_abi_forward.c — the C ABI of libdsen2_hip.so used from plain C: no Python, no torch, only include/dsen2_hip.h and the HIP
 * runtime API for device memory.  What a non-Python host (or the cgo / JNI stub of another runtime) does:
 *
 *   dsen2_model_create  ->  dsen2_model_load_weights  ->  dsen2_model_workspace_bytes  ->  dsen2_model_forward
 *
 * Build and run (tests/test_gpu_c_abi_example.py does exactly this and compares the output file, bit for bit, with the
 * Python host's result for the same weights and inputs):
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_forward.c -Ldsen2_amd -ldsen2_hip \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/dsen2_amd -Wl,-rpath,/opt/rocm/lib -o build/c_abi_forward
 *   build/c_abi_forward weights.f32 x10.f32 x20.f32 out.f32 <n> <h> <w> <num_layers> <feature_size> <precision>
 * The four files are raw little-endian float32: keras-flat weights (dsen2_model_num_params values), x10 [n,4,h,w],
 * x20 [n,6,h,w], out [n,6,h,w].
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "dsen2_hip.h"

static float *read_f32(const char *path, size_t count) {
  float *buf = (float *)malloc(count * sizeof(float));
  FILE *f = fopen(path, "rb");
  if (!buf || !f || fread(buf, sizeof(float), count, f) != count) {
    fprintf(stderr, "cannot read %zu floats from %s\n", count, path);
    exit(2);
  }
  fclose(f);
  return buf;
}

#define HIP_OK(expr)                                                             \
  do {                                                                           \
    hipError_t e_ = (expr);                                                      \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_));                 \
      return 3;                                                                  \
    }                                                                            \
  } while (0)
#define DSEN2_CHECK(expr)                                                        \
  do {                                                                           \
    int rc_ = (expr);                                                            \
    if (rc_ != DSEN2_OK) {                                                       \
      fprintf(stderr, "%s: error %d: %s\n", #expr, rc_, dsen2_last_error());     \
      return 4;                                                                  \
    }                                                                            \
  } while (0)

int main(int argc, char **argv) {
  if (argc != 11) {
    fprintf(stderr, "usage: %s weights.f32 x10.f32 x20.f32 out.f32 n h w num_layers feature_size precision\n", argv[0]);
    return 1;
  }
  const int n = atoi(argv[5]), h = atoi(argv[6]), w = atoi(argv[7]);
  const int d = atoi(argv[8]), feat = atoi(argv[9]), precision = atoi(argv[10]);
  printf("%s, %d gfx950 device(s)\n", dsen2_version(), dsen2_device_count());

  dsen2_model *m = NULL;
  DSEN2_CHECK(dsen2_model_create(&m, 4, 6, 0, d, feat, precision));      /* s2model(((4,.,.),(6,.,.)), d, feat) */
  const size_t n_params = dsen2_model_num_params(m);
  float *weights = read_f32(argv[1], n_params);
  DSEN2_CHECK(dsen2_model_load_weights(m, weights, n_params));           /* model.load_weights(...) */

  const size_t pix = (size_t)n * h * w;
  float *x10 = read_f32(argv[2], pix * 4), *x20 = read_f32(argv[3], pix * 6);
  float *d10 = NULL, *d20 = NULL, *dout = NULL;
  void *ws = NULL;
  size_t ws_bytes = 0;
  DSEN2_CHECK(dsen2_model_workspace_bytes(m, n, h, w, &ws_bytes));
  HIP_OK(hipMalloc((void **)&d10, pix * 4 * sizeof(float)));
  HIP_OK(hipMalloc((void **)&d20, pix * 6 * sizeof(float)));
  HIP_OK(hipMalloc((void **)&dout, pix * 6 * sizeof(float)));
  HIP_OK(hipMalloc(&ws, ws_bytes));
  HIP_OK(hipMemcpy(d10, x10, pix * 4 * sizeof(float), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d20, x20, pix * 6 * sizeof(float), hipMemcpyHostToDevice));

  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  DSEN2_CHECK(dsen2_model_forward(m, d10, d20, NULL, dout, n, h, w, ws, ws_bytes, stream));   /* model.predict([p10, p20]) */
  HIP_OK(hipStreamSynchronize(stream));

  float *out = (float *)malloc(pix * 6 * sizeof(float));
  HIP_OK(hipMemcpy(out, dout, pix * 6 * sizeof(float), hipMemcpyDeviceToHost));
  FILE *f = fopen(argv[4], "wb");
  if (!f || fwrite(out, sizeof(float), pix * 6, f) != pix * 6) {
    fprintf(stderr, "cannot write %s\n", argv[4]);
    return 5;
  }
  fclose(f);
  double sum = 0.0;
  for (size_t i = 0; i < pix * 6; ++i) sum += out[i];
  printf("forward of %d patches of %dx%d (d=%d, F=%d, precision %d): %zu parameters, workspace %zu bytes, output sum %.6f\n",
         n, h, w, d, feat, precision, n_params, ws_bytes, sum);

  /* the error contract: a forward that cannot run says why instead of crashing */
  if (dsen2_model_forward(m, d10, d20, NULL, dout, n, h, w, ws, ws_bytes / 2, stream) != DSEN2_ERR_WORKSPACE) {
    fprintf(stderr, "expected DSEN2_ERR_WORKSPACE for a short workspace\n");
    return 6;
  }
  printf("short workspace refused: %s\n", dsen2_last_error());

  dsen2_model_destroy(m);
  HIP_OK(hipStreamDestroy(stream));
  HIP_OK(hipFree(d10)); HIP_OK(hipFree(d20)); HIP_OK(hipFree(dout)); HIP_OK(hipFree(ws));
  free(weights); free(x10); free(x20); free(out);
  return 0;
}
