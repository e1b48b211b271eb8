// patch_ops.hip — the HBM-bound kernels either side of the network: input concat/pack, mirror-bilinear
// up-sampling, overlapped tiling and recomposition (utils/patches.py of the reference).
// All of them are pure gathers: one thread per OUTPUT element, so there are no write races and every
// store is coalesced; reads are served by L2 (each source line is touched by a handful of neighbours).
#include "dsen2_internal.h"

namespace dsen2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline unsigned grid_for(size_t work, int block) {
  size_t g = (work + block - 1) / block;
  const size_t cap = 256 * 16;   // 256 CUs x 16 blocks, grid-stride the rest
  return (unsigned)(g < cap ? (g ? g : 1) : cap);
}

// ---- concat + NCHW -> NHWC16 ------------------------------------------------------------------
// keras Concatenate(axis=1) of [input10, input20(, input60)] (utils/DSen2Net.py:24,26) folded into the
// layout change the first convolution needs: out[n][y][x][0..15] = (x10 | x20 | x60 | zeros).
__global__ __launch_bounds__(256) void pack_inputs_kernel(const float* __restrict__ x10, const float* __restrict__ x20,
                                                          const float* __restrict__ x60, int c10, int c20, int c60,
                                                          float* __restrict__ out, size_t npix_total, size_t plane) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix_total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / plane, q = i - n * plane;
    float v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      float t = 0.f;
      if (c < c10) t = x10[(n * c10 + c) * plane + q];
      else if (c < c10 + c20) t = x20[(n * c20 + (c - c10)) * plane + q];
      else if (c < c10 + c20 + c60) t = x60[(n * c60 + (c - c10 - c20)) * plane + q];
      v[c] = t;
    }
    f32x4* dst = reinterpret_cast<f32x4*>(out + i * 16);
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[k] = f32x4{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
  }
}

hipError_t launch_pack_inputs(const float* x10, const float* x20, const float* x60, int c10, int c20, int c60,
                              float* out, int n, int h, int w, hipStream_t stream) {
  const size_t plane = (size_t)h * w, total = plane * n;
  hipLaunchKernelGGL(pack_inputs_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, x10, x20, x60, c10, c20,
                     c60, out, total, plane);
  return hipGetLastError();
}

// ---- mirror-bilinear up-sampling (interp_patches, utils/patches.py:11-16) ----------------------
// skimage.transform.resize(x/30000, (oh,ow), mode='reflect')*30000, order 1:
//   src = scale*dst + offset with scale = in/out, offset = 0.5*scale - 0.5, evaluated in float32 with a
//   separate multiply and add exactly as skimage 0.18.3 does (warp() casts its matrix to the image dtype);
//   neighbours floor(src)/ceil(src), folded back by mirroring WITHOUT repeating the edge sample;
//   the blend follows skimage's compiled `bilinear_interpolation` for a float32 image operation by operation:
//     top = (1.0 - double(dc)) * double(left) + double(float32(dc * right))      (`dc * right`: both float32 in the Cython
//     source, so that product alone is a float32 multiply; the literal 1 promotes everything else to double)
//     out = float32((1.0 - double(dr)) * top + double(dr) * bottom)
//   — the captured outputs of the reference's interp_patches are reproduced BIT FOR BIT (tests/test_gpu_patches.py).  warp()'s
//   clip to the input's [min, max] is a no-op for this arithmetic (a plateau of equal taps returns its value exactly, every
//   operation is monotone in the taps; the tests' CPU restatement keeps the clip and agrees) and is not computed.
__device__ __forceinline__ int mirror_index(int i, int dim) {
  if (dim == 1) return 0;
  const int cmax = dim - 1;
  if (i < 0) {
    const int k = -i;
    return ((k / cmax) & 1) ? cmax - (k % cmax) : (k % cmax);
  }
  if (i > cmax) return ((i / cmax) & 1) ? cmax - (i % cmax) : (i % cmax);
  return i;
}

// mirror_index for an index at most one image away (-cmax <= i <= 2 * cmax: one reflection, no division, no branch) —
// every tap of an up-sampler's window; images of one or two pixels (several reflections) take the general form.
__device__ __forceinline__ int mirror_index_near(int i, int dim) {
  const int cmax = dim - 1;
  int j = i < 0 ? -i : i;
  j = j > cmax ? 2 * cmax - j : j;
  if (__builtin_expect((unsigned)j > (unsigned)cmax, 0)) j = mirror_index(i, dim);
  return j;
}

// One thread = 4 consecutive output columns x kUpRows consecutive output rows of one plane.  The reference blends
// horizontally first (top = row r0, bot = row r1, same formula) and then vertically, so the horizontal blend H[r][oj] of
// an input row is the same number whichever output row asks for it: a thread keeps the two most recent H rows (an
// up-sampler's consecutive output rows use (k, k+1), (k, k+1), (k+1, k+2), ...) and, inside a row, a source column's
// quotient by 30000 once per four outputs.  Same operations on the same values as the one-output-at-a-time form:
// the results are its bits (checked against the previous library on ragged shapes, tools/ history; the golden
// vectors of tests/test_gpu_patches.py pin both against skimage).
constexpr int kUpRows = 8;

__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                       size_t total_threads, int h, int w, int oh, int ow, int gpr,
                                                       int row_blocks, float sy, float oy, float sx, float ox,
                                                       float post_div) {
  const size_t per_plane = (size_t)row_blocks * gpr;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_threads; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i / per_plane;
    const int rem = (int)(i - p * per_plane);
    const int rb = rem / gpr, oj0 = (rem - rb * gpr) * 4;
    const float* const plane = in + p * (size_t)h * w;
    // column taps of the four outputs (identical for every row)
    int c0[4], c1[4];
    double dc[4];
    float dcf[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int oj = oj0 + e < ow ? oj0 + e : ow - 1;
      const float c = __fadd_rn(__fmul_rn(sx, (float)oj), ox);
      const float cf = floorf(c);
      c0[e] = mirror_index((int)cf, w);
      c1[e] = mirror_index((int)ceilf(c), w);
      dcf[e] = __fsub_rn(c, cf);
      dc[e] = (double)dcf[e];
    }
    // horizontal blend of input row r at the four output columns
    auto hrow = [&](int r, double (&H)[4]) {
      const float* const src = plane + (size_t)r * w;
      int pc0 = -1, pc1 = -1;
      double pq0 = 0.0, pq1 = 0.0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double q0, q1;
        if (c0[e] == pc0) q0 = pq0;
        else if (c0[e] == pc1) q0 = pq1;
        else q0 = (double)__fdiv_rn(src[c0[e]], 30000.0f);
        if (c1[e] == c0[e]) q1 = q0;
        else if (c1[e] == pc1) q1 = pq1;
        else if (c1[e] == pc0) q1 = pq0;
        else q1 = (double)__fdiv_rn(src[c1[e]], 30000.0f);
        pc0 = c0[e]; pc1 = c1[e]; pq0 = q0; pq1 = q1;
        H[e] = (1.0 - dc[e]) * q0 + (double)__fmul_rn(dcf[e], (float)q1);      // (q1 is a float32 quotient: the cast is exact)
      }
    };
    int ia = -1, ib = -1;                       // input rows held in Ha / Hb
    double Ha[4], Hb[4];
    bool a_older = true;                        // which of the two is replaced next
    for (int k = 0; k < kUpRows; ++k) {
      const int oi = rb * kUpRows + k;
      if (oi >= oh) break;
      const float r = __fadd_rn(__fmul_rn(sy, (float)oi), oy);
      const float rf = floorf(r);
      const int r0 = mirror_index((int)rf, h), r1 = mirror_index((int)ceilf(r), h);
      const double dr = (double)__fsub_rn(r, rf);
      // make both rows resident (never evicting the one the other tap needs)
      if (r0 != ia && r0 != ib) {
        if (a_older && ia != r1) { hrow(r0, Ha); ia = r0; a_older = false; }
        else if (ib != r1) { hrow(r0, Hb); ib = r0; a_older = true; }
        else { hrow(r0, Ha); ia = r0; a_older = false; }
      }
      if (r1 != ia && r1 != ib) {
        if (ia != r0) { hrow(r1, Ha); ia = r1; a_older = false; }
        else { hrow(r1, Hb); ib = r1; a_older = true; }
      }
      float res[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double top = r0 == ia ? Ha[e] : Hb[e];
        const double bot = r1 == ia ? Ha[e] : Hb[e];
        const float v = __fmul_rn((float)((1.0 - dr) * top + dr * bot), 30000.0f);
        res[e] = post_div == 1.0f ? v : __fdiv_rn(v, post_div);   // folds `p20 /= SCALE` (testing/supres.py:24)
      }
      float* dst = out + (p * oh + oi) * (size_t)ow + oj0;
      if ((ow & 3) == 0) {
        *reinterpret_cast<f32x4*>(dst) = f32x4{res[0], res[1], res[2], res[3]};
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (oj0 + e < ow) dst[e] = res[e];
      }
    }
  }
}

// The same arithmetic without a branch in sight, for up-sampling by 2 or more in both directions (every use in this
// package: x2 and x6).  upsample_kernel above fetches a source sample only when its reuse chain misses — every load sits
// behind a branch, so a thread's ~20 loads are issued one after the other and it waits for each.  Here a thread's four output
// columns read from a window of 4 consecutive source columns (3 * sx + 1 < 4 for sx <= 1/2) and its eight output rows from NR
// consecutive source rows (7 * sy + 2 <= NR): all NR x 4 samples are loaded at once, divided, blended horizontally
// (operands picked by index selects, the same two products — one double, one float32 — and one sum per output column as above) and parked in LDS
// ([row][column][thread]: a thread only ever reads its own values, no barrier); the vertical blend then indexes LDS by row.
// Same operations on the same values as upsample_kernel: bit-identical (tests/test_gpu_patches.py against
// dsen2_upsample_mirror_bilinear_ref; tools/ab_upsample_bits.py against the previous library), 1.3-1.4x faster.
template <int NR>
__global__ __launch_bounds__(256) void upsample_window_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                              size_t total_threads, int h, int w, int oh, int ow, int gpr,
                                                              int row_blocks, float sy, float oy, float sx, float ox,
                                                              float post_div) {
  __shared__ double h_s[NR * 4 * 256];
  const int tid = threadIdx.x;
  const size_t per_plane = (size_t)row_blocks * gpr;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + tid; i < total_threads; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i / per_plane;
    const int rem = (int)(i - p * per_plane);
    const int rb = rem / gpr, oj0 = (rem - rb * gpr) * 4;
    const float* const plane = in + p * (size_t)h * w;
    // columns: the window starts at the first output's left tap
    int i0[4], i1[4];
    double dc[4];
    float dcf[4];
    int cb = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int oj = oj0 + e < ow ? oj0 + e : ow - 1;
      const float c = __fadd_rn(__fmul_rn(sx, (float)oj), ox);
      const float cf = floorf(c);
      if (e == 0) cb = (int)cf;
      i0[e] = (int)cf - cb;
      i1[e] = (int)ceilf(c) - cb;
      dcf[e] = __fsub_rn(c, cf);
      dc[e] = (double)dcf[e];
    }
    int wcol[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wcol[j] = mirror_index_near(cb + j, w);
    // rows: the window starts at the first output row's upper tap
    const int oi0 = rb * kUpRows;
    const int ru0 = (int)floorf(__fadd_rn(__fmul_rn(sy, (float)oi0), oy));
    float q[NR][4];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const float* const src = plane + (size_t)mirror_index_near(ru0 + j, h) * w;
#pragma unroll
      for (int c = 0; c < 4; ++c) q[j][c] = src[wcol[c]];
    }
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) q[j][c] = __fdiv_rn(q[j][c], 30000.0f);
    auto pick = [](const float (&v)[4], int k) -> float {
      const float lo = k == 1 ? v[1] : v[0], hi = k == 3 ? v[3] : v[2];
      return k >= 2 ? hi : lo;
    };
#pragma unroll
    for (int j = 0; j < NR; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double q0 = (double)pick(q[j], i0[e]);
        h_s[(j * 4 + e) * 256 + tid] = (1.0 - dc[e]) * q0 + (double)__fmul_rn(dcf[e], pick(q[j], i1[e]));
      }
#pragma unroll
    for (int k = 0; k < kUpRows; ++k) {
      const int oi = oi0 + k;
      if (oi >= oh) break;
      const float r = __fadd_rn(__fmul_rn(sy, (float)oi), oy);
      const float rf = floorf(r);
      const int j0 = (int)rf - ru0, j1 = (int)ceilf(r) - ru0;
      const double dr = (double)__fsub_rn(r, rf);
      float res[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double top = h_s[(j0 * 4 + e) * 256 + tid], bot = h_s[(j1 * 4 + e) * 256 + tid];
        const float v = __fmul_rn((float)((1.0 - dr) * top + dr * bot), 30000.0f);
        res[e] = post_div == 1.0f ? v : __fdiv_rn(v, post_div);   // folds `p20 /= SCALE` (testing/supres.py:24)
      }
      float* dst = out + (p * oh + oi) * (size_t)ow + oj0;
      if ((ow & 3) == 0) {
        *reinterpret_cast<f32x4*>(dst) = f32x4{res[0], res[1], res[2], res[3]};
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (oj0 + e < ow) dst[e] = res[e];
      }
    }
  }
}

hipError_t launch_upsample(const float* in, float* out, int planes, int h, int w, int oh, int ow, float post_div,
                           hipStream_t stream, bool general) {
  const int gpr = (ow + 3) / 4;                       // 4-column groups per output row
  const int row_blocks = (oh + kUpRows - 1) / kUpRows;
  const size_t total = (size_t)planes * row_blocks * gpr;
  const double fy = (double)h / oh, fx = (double)w / ow;
  const float sy = (float)fy, oy = (float)(0.5 * fy - 0.5), sx = (float)fx, ox = (float)(0.5 * fx - 0.5);
  const dim3 grid(grid_for(total, 256)), block(256);
  // the windowed form needs its taps inside the window: 3 * sx + 1 < 4 columns, 7 * sy + 2 <= NR rows (float32 coordinates:
  // keep a margin)
  if (!general && fx <= 0.5 && fy <= 0.5) {
    if (7.0 * fy + 2.0 < 3.99)
      hipLaunchKernelGGL(upsample_window_kernel<4>, grid, block, 0, stream, in, out, total, h, w, oh, ow, gpr, row_blocks, sy, oy, sx, ox, post_div);
    else
      hipLaunchKernelGGL(upsample_window_kernel<6>, grid, block, 0, stream, in, out, total, h, w, oh, ow, gpr, row_blocks, sy, oy, sx, ox, post_div);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(upsample_kernel, grid, block, 0, stream, in, out, total, h, w, oh, ow, gpr, row_blocks, sy, oy, sx, ox, post_div);
  return hipGetLastError();
}

// ---- overlapped tiling (get_test_patches{,60}: np.pad 'symmetric' + crop + HWC->CHW) -----------
// numpy 'symmetric' pad repeats the edge sample: padded index q maps to |q - b| - (q < b) ... written out:
__device__ __forceinline__ int symmetric_index(int q, int n) {   // q = index in the image frame, may be <0 or >=n
  if (q < 0) q = -1 - q;
  if (q >= n) q = 2 * n - 1 - q;
  // One reflection covers every origin the tiling arithmetic produces (|pad| <= border <= n).  An origin that does
  // not belong to this image (inconsistent image sizes handed in by a caller; the Python layer raises ValueError
  // first) must still never read outside it: clamp.
  return q < 0 ? 0 : (q >= n ? n - 1 : q);
}

// One thread per output PIXEL: it reads the pixel's C interleaved source values once (contiguous 4*C bytes) and
// writes them to the C output planes; each plane's stores are coalesced across the threads of a row.
template <int C>
__global__ __launch_bounds__(256) void tile_gather_kernel(const float* __restrict__ img, int H, int W, int border,
                                                          const int* __restrict__ origins, size_t total_pix, int P,
                                                          float divisor, float* __restrict__ patches) {
  const size_t pp = (size_t)P * P;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_pix; i += (size_t)gridDim.x * blockDim.x) {
    const size_t k = i / pp;
    const int r2 = (int)(i - k * pp);
    const int y = r2 / P, x = r2 - y * P;
    const int yy = symmetric_index(origins[2 * k] + y - border, H);
    const int xx = symmetric_index(origins[2 * k + 1] + x - border, W);
    const float* src = img + ((size_t)yy * W + xx) * C;
    float* dst = patches + k * pp * C + r2;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float v = src[c];
      dst[(size_t)c * pp] = divisor == 1.0f ? v : __fdiv_rn(v, divisor);   // folds `p10 /= SCALE` (supres.py:23)
    }
  }
}

__global__ __launch_bounds__(256) void tile_gather_generic_kernel(const float* __restrict__ img, int H, int W, int C,
                                                                  int border, const int* __restrict__ origins,
                                                                  size_t total, int P, float divisor,
                                                                  float* __restrict__ patches) {
  const size_t pp = (size_t)P * P, per_patch = pp * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t k = i / per_patch;
    const int rem = (int)(i - k * per_patch);
    const int c = rem / (int)pp, r2 = rem - c * (int)pp;
    const int y = r2 / P, x = r2 - y * P;
    const int yy = symmetric_index(origins[2 * k] + y - border, H);
    const int xx = symmetric_index(origins[2 * k + 1] + x - border, W);
    const float v = img[((size_t)yy * W + xx) * C + c];
    patches[i] = divisor == 1.0f ? v : __fdiv_rn(v, divisor);
  }
}

hipError_t launch_tile_gather(const float* img, int H, int W, int C, int border, const int* origins, int count,
                              int P, float divisor, float* patches, hipStream_t stream) {
  const size_t total_pix = (size_t)count * P * P;
  if (total_pix == 0) return hipSuccess;
  const dim3 grid(grid_for(total_pix, 256)), block(256);
  switch (C) {      // the Sentinel-2 band groups: 4 (10 m), 6 (20 m), 2 (60 m)
    case 2: hipLaunchKernelGGL(tile_gather_kernel<2>, grid, block, 0, stream, img, H, W, border, origins, total_pix, P, divisor, patches); break;
    case 4: hipLaunchKernelGGL(tile_gather_kernel<4>, grid, block, 0, stream, img, H, W, border, origins, total_pix, P, divisor, patches); break;
    case 6: hipLaunchKernelGGL(tile_gather_kernel<6>, grid, block, 0, stream, img, H, W, border, origins, total_pix, P, divisor, patches); break;
    default:
      hipLaunchKernelGGL(tile_gather_generic_kernel, dim3(grid_for(total_pix * C, 256)), block, 0, stream, img, H, W, C,
                         border, origins, total_pix * C, P, divisor, patches);
  }
  return hipGetLastError();
}

// ---- recomposition (recompose_images, utils/patches.py:374-405) --------------------------------
// One thread per output PIXEL: C coalesced plane reads (neighbouring threads = neighbouring x of the same patch
// row), one contiguous 4*C-byte HWC write.
template <int C>
__global__ __launch_bounds__(256) void recompose_kernel(const float* __restrict__ patches, int P, int border,
                                                        float* __restrict__ img, int H, int W, int x_tiles,
                                                        int y_tiles, float scale, size_t first_pix, size_t total_pix, int c_rt) {
  const int inner = P - 2 * border;
  const int cc = C > 0 ? C : c_rt;
  const size_t pp = (size_t)P * P;
  // pixels [first_pix, total_pix) of the image: a band of rows (the whole image: first_pix = 0)
  for (size_t i = first_pix + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_pix; i += (size_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
    // the LAST tile covering (y, x) wins, as in the reference's sequential overwrite
    const int ty = (y >= H - inner) ? y_tiles - 1 : y / inner;
    const int tx = (x >= W - inner) ? x_tiles - 1 : x / inner;
    const int ys = (ty == y_tiles - 1) ? H - inner : ty * inner;
    const int xs = (tx == x_tiles - 1) ? W - inner : tx * inner;
    const size_t k = (size_t)ty * x_tiles + tx;
    const float* src = patches + k * cc * pp + (size_t)(border + y - ys) * P + (border + x - xs);
    float* dst = img + i * cc;
    if constexpr (C > 0) {
#pragma unroll
      for (int c = 0; c < C; ++c) dst[c] = src[(size_t)c * pp] * scale;
    } else {
      for (int c = 0; c < cc; ++c) dst[c] = src[(size_t)c * pp] * scale;
    }
  }
}

// rows [row0, row1) of the image only (the whole image: 0, H).  The patches those rows read must have been written; the
// others are not touched.
hipError_t launch_recompose(const float* patches, int count, int C, int P, int border, float* img, int H, int W,
                            float scale, int row0, int row1, hipStream_t stream) {
  const int inner = P - 2 * border;
  if (inner <= 0 || H < inner || W < inner) return hipErrorInvalidValue;
  const int x_tiles = (W + inner - 1) / inner, y_tiles = (H + inner - 1) / inner;
  if ((long long)x_tiles * y_tiles > count) return hipErrorInvalidValue;
  if (row0 < 0 || row1 > H || row0 > row1) return hipErrorInvalidValue;
  if (row0 == row1) return hipSuccess;
  const size_t first_pix = (size_t)row0 * W, total_pix = (size_t)row1 * W;
  const dim3 grid(grid_for(total_pix - first_pix, 256)), block(256);
  switch (C) {
    case 2: hipLaunchKernelGGL(recompose_kernel<2>, grid, block, 0, stream, patches, P, border, img, H, W, x_tiles, y_tiles, scale, first_pix, total_pix, C); break;
    case 6: hipLaunchKernelGGL(recompose_kernel<6>, grid, block, 0, stream, patches, P, border, img, H, W, x_tiles, y_tiles, scale, first_pix, total_pix, C); break;
    default: hipLaunchKernelGGL(recompose_kernel<0>, grid, block, 0, stream, patches, P, border, img, H, W, x_tiles, y_tiles, scale, first_pix, total_pix, C);
  }
  return hipGetLastError();
}

}  // namespace dsen2
