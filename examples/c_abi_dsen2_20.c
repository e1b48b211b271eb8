/* c_abi_dsen2_20.c — the WHOLE drop-in call DSen2_20(d10, d20) (testing/supres.py:15-30) from plain C: no Python, no torch.
 * Everything the Python host (dsen2_amd/supres.py) does is a few integers of tiling arithmetic plus calls into the C ABI:
 *
 *   origins (host)          patches.py:45-60: stride = P/2 - 2*(border/2) low-resolution pixels, one extra clamped origin when
 *                           the extent is not a multiple of it, row-major
 *   dsen2_tile_gather x 2   np.pad 'symmetric' + crop + HWC->CHW; `p10 /= SCALE` folded         (patches.py:27-28,58-72)
 *   dsen2_upsample_mirror_bilinear   interp_patches; `p20 /= SCALE` folded                         (patches.py:11-16)
 *   dsen2_model_forward     s2model(...).predict                                                   (DSen2Net.py:18-43)
 *   dsen2_recompose         recompose_images; `images *= SCALE` folded                             (patches.py:374-405)
 *
 * tests/test_c_abi_example.py builds it with gcc -std=c99 -Wall -Werror and compares its output file with
 * dsen2_amd.supres.DSen2_20 for the same rasters and weights, bit for bit.
 *   build/c_abi_dsen2_20 weights.f32 d10.f32 d20.f32 out.f32 <x> <y> <num_layers> <feature_size> <precision>
 * d10 [x,y,4], d20 [x/2,y/2,6], out [x,y,6]: raw little-endian float32, HWC.  One batch: size the image to fit the GPU.
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "dsen2_hip.h"

#define PATCH 128
#define BORDER 8
#define SCALE 2000.0f

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

/* Origins along one axis of the low-resolution image, in PADDED coordinates (patches.py:45-53): 0, stride, ... and, when the
 * extent is not a multiple of the stride, the clamped origin extent + 2*border - patch. */
static int axis_origins(int extent, int patch, int border, int *out) {
  const int stride = patch - 2 * border;
  int k = 0;
  for (int i = 0; i < extent / stride; ++i) out[k++] = i * stride;
  if (extent % stride != 0) out[k++] = extent + 2 * border - patch;
  return k;
}

int main(int argc, char **argv) {
  if (argc != 10) {
    fprintf(stderr, "usage: %s weights.f32 d10.f32 d20.f32 out.f32 x y num_layers feature_size precision\n", argv[0]);
    return 1;
  }
  const int X = atoi(argv[5]), Y = atoi(argv[6]);
  const int d = atoi(argv[7]), feat = atoi(argv[8]), precision = atoi(argv[9]);
  const int x2 = X / 2, y2 = Y / 2, p_lr = PATCH / 2, b_lr = BORDER / 2;
  if (X % 2 || Y % 2 || x2 + 2 * b_lr < p_lr || y2 + 2 * b_lr < p_lr) {
    fprintf(stderr, "the 10 m image must be even-sized and hold at least one patch\n");
    return 1;
  }
  /* tiling arithmetic */
  int *oi = (int *)malloc(sizeof(int) * (size_t)(x2 + 2)), *oj = (int *)malloc(sizeof(int) * (size_t)(y2 + 2));
  const int ni = axis_origins(x2, p_lr, b_lr, oi), nj = axis_origins(y2, p_lr, b_lr, oj);
  const int count = ni * nj;
  int *org20 = (int *)malloc(sizeof(int) * 2 * (size_t)count), *org10 = (int *)malloc(sizeof(int) * 2 * (size_t)count);
  for (int i = 0; i < ni; ++i)
    for (int j = 0; j < nj; ++j) {
      const int k = i * nj + j;
      org20[2 * k] = oi[i]; org20[2 * k + 1] = oj[j];
      org10[2 * k] = 2 * oi[i]; org10[2 * k + 1] = 2 * oj[j];           /* HR crop = 2 x LR crop (patches.py:67) */
    }
  printf("%s: %d x %d -> %d patches of %d\n", dsen2_version(), X, Y, count, PATCH);

  dsen2_model *m = NULL;
  DSEN2_CHECK(dsen2_model_create(&m, 4, 6, 0, d, feat, precision));
  float *weights = read_f32(argv[1], dsen2_model_num_params(m));
  DSEN2_CHECK(dsen2_model_load_weights(m, weights, dsen2_model_num_params(m)));
  float *h10 = read_f32(argv[2], (size_t)X * Y * 4), *h20 = read_f32(argv[3], (size_t)x2 * y2 * 6);

  const size_t pp = (size_t)PATCH * PATCH, plr = (size_t)p_lr * p_lr;
  float *d10, *d20, *p10, *p20lr, *p20, *pred, *img;
  int *dorg10, *dorg20;
  void *ws;
  size_t ws_bytes = 0;
  DSEN2_CHECK(dsen2_model_workspace_bytes(m, count, PATCH, PATCH, &ws_bytes));
  HIP_OK(hipMalloc((void **)&d10, sizeof(float) * (size_t)X * Y * 4));
  HIP_OK(hipMalloc((void **)&d20, sizeof(float) * (size_t)x2 * y2 * 6));
  HIP_OK(hipMalloc((void **)&p10, sizeof(float) * count * 4 * pp));
  HIP_OK(hipMalloc((void **)&p20lr, sizeof(float) * count * 6 * plr));
  HIP_OK(hipMalloc((void **)&p20, sizeof(float) * count * 6 * pp));
  HIP_OK(hipMalloc((void **)&pred, sizeof(float) * count * 6 * pp));
  HIP_OK(hipMalloc((void **)&img, sizeof(float) * (size_t)X * Y * 6));
  HIP_OK(hipMalloc((void **)&dorg10, sizeof(int) * 2 * (size_t)count));
  HIP_OK(hipMalloc((void **)&dorg20, sizeof(int) * 2 * (size_t)count));
  HIP_OK(hipMalloc(&ws, ws_bytes));
  HIP_OK(hipMemcpy(d10, h10, sizeof(float) * (size_t)X * Y * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d20, h20, sizeof(float) * (size_t)x2 * y2 * 6, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dorg10, org10, sizeof(int) * 2 * (size_t)count, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dorg20, org20, sizeof(int) * 2 * (size_t)count, hipMemcpyHostToDevice));

  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  DSEN2_CHECK(dsen2_tile_gather(d10, X, Y, 4, BORDER, dorg10, count, PATCH, SCALE, p10, stream));
  DSEN2_CHECK(dsen2_tile_gather(d20, x2, y2, 6, b_lr, dorg20, count, p_lr, 1.0f, p20lr, stream));
  DSEN2_CHECK(dsen2_upsample_mirror_bilinear(p20lr, p20, count * 6, p_lr, p_lr, PATCH, PATCH, SCALE, stream));
  DSEN2_CHECK(dsen2_model_forward(m, p10, p20, NULL, pred, count, PATCH, PATCH, ws, ws_bytes, stream));
  if ((x2 / (p_lr - 2 * b_lr) + 1) * (y2 / (p_lr - 2 * b_lr) + 1) == 1) {      /* the reference ALLOCATES (k_i + 1)(k_j + 1) patches */
    fprintf(stderr, "a single allocated patch is returned uncropped by recompose_images (patches.py:375-376): not covered here\n");
    return 1;
  }
  DSEN2_CHECK(dsen2_recompose(pred, count, 6, PATCH, BORDER, img, X, Y, SCALE, stream));
  HIP_OK(hipStreamSynchronize(stream));

  float *out = (float *)malloc(sizeof(float) * (size_t)X * Y * 6);
  HIP_OK(hipMemcpy(out, img, sizeof(float) * (size_t)X * Y * 6, hipMemcpyDeviceToHost));
  FILE *f = fopen(argv[4], "wb");
  if (!f || fwrite(out, sizeof(float), (size_t)X * Y * 6, f) != (size_t)X * Y * 6) {
    fprintf(stderr, "cannot write %s\n", argv[4]);
    return 2;
  }
  fclose(f);
  dsen2_model_destroy(m);
  HIP_OK(hipStreamDestroy(stream));
  printf("wrote %s [%d, %d, 6]\n", argv[4], X, Y);
  return 0;
}
