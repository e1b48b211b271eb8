// conv3x3_out.hip — the network's last convolution: F -> Cout (6 or 2 channels) 3x3 'same' + bias + the
// low-resolution skip input, NHWC in, NCHW out (utils/DSen2Net.py:35,38,41).
//
// With only 6 (or 2) output channels a 32-wide MFMA block would be 81 % (94 %) padding, so this kernel uses
// v_mfma_f32_16x16x4_f32 (exact f32, 32 cycles): M = 16 pixels (one tile row), N = 16 output channels (Cout
// zero-padded), k = 4.  A lane fetches 4 consecutive input channels with one ds_read_b128 and feeds 4 MFMAs;
// MFMA j of a 16-channel step contracts channels {16s + 4q + j : q = lane>>4} on both operands.
// D[px][o] puts output channel o on lane&15 and 4 consecutive pixels in a lane's 4 registers, which is exactly
// 16 contiguous bytes of an NCHW row: the epilogue is one float4 load (skip) and one float4 store per block.
//
// One workgroup = 8 waves = one 16x16 tile; wave w owns tile rows 2w and 2w+1.  LDS: halo tile of 16 channels
// (double buffered) + the 9 taps x 16 channels x 16 outputs of the current channel chunk (double buffered);
// one barrier per 16-channel chunk (72 MFMAs per wave); two workgroups per CU.
#include "dsen2_internal.h"

namespace dsen2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace outk {
constexpr int KC = 16;                         // 16-channel chunks: 70 KB of LDS = two workgroups per CU, one's chunk hand-over
                                               // under the other's MFMAs (32-channel chunks, one workgroup per CU: 205 -> 171 us)
constexpr int NO = 16;                         // padded output channels
constexpr int THREADS = 512;
constexpr int PSTR = KC + 4;
constexpr int IN_FLOATS = kHaloPix * PSTR;
constexpr int WCH = 9 * KC * NO;               // floats per channel chunk of weights (all 9 taps)
constexpr int IN_PIECES = kHaloPix * (KC / 4);
constexpr int IN_ROUNDS = (IN_PIECES + THREADS - 1) / THREADS;
constexpr int W_PIECES = WCH / 4;
constexpr int W_ROUNDS = (W_PIECES + THREADS - 1) / THREADS;
constexpr size_t LDS_BYTES = (size_t)(2 * IN_FLOATS + 2 * WCH) * sizeof(float);   // 70,272 B
}  // namespace outk

template <int CIN>
__global__ __launch_bounds__(outk::THREADS, 2) void conv3x3_out_kernel(const ConvParams p) {
  using namespace outk;
  constexpr int NCC = CIN / KC;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const in_s = smem;                    // [2][324][PSTR]
  float* const w_s = smem + 2 * IN_FLOATS;     // [2][9][KC/4][NO][4]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;
  const int q = lane >> 4;

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const int img = lid / tiles_per_img;
  const int trem = lid - img * tiles_per_img;
  const int tyi = trem / p.tiles_x;
  const int ty0 = tyi * kTile, tx0 = (trem - tyi * p.tiles_x) * kTile;
  const size_t img_pix = (size_t)p.h * p.w;
  const float* const in_img = p.in + (size_t)img * img_pix * CIN;

  int g_off[IN_ROUNDS], s_off[IN_ROUNDS];
#pragma unroll
  for (int r = 0; r < IN_ROUNDS; ++r) {
    const int piece = r * THREADS + tid;
    const int hp = piece / (KC / 4), qq = piece - hp * (KC / 4);
    const int hy = hp / kHalo, hx = hp - hy * kHalo;
    const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
    const bool have = piece < IN_PIECES;
    const bool inb = have && (unsigned)gy < (unsigned)p.h && (unsigned)gx < (unsigned)p.w;
    s_off[r] = have ? hp * PSTR + qq * 4 : -1;
    g_off[r] = inb ? (gy * p.w + gx) * CIN + qq * 4 : -1;
  }
  auto load_in = [&](int r, int cc) -> f32x4 {     // branch-free: a clamped address, zeros selected at the LDS store
    return *reinterpret_cast<const f32x4*>(in_img + (g_off[r] >= 0 ? g_off[r] : 0) + cc * KC);
  };
  auto store_in = [&](float* buf, int r, f32x4 t) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = g_off[r] >= 0 ? t[e] : 0.f;
    if (s_off[r] >= 0) *reinterpret_cast<f32x4*>(buf + s_off[r]) = v;
  };
  auto load_w = [&](int r, int cc) -> f32x4 {
    const int piece = r * THREADS + tid;
    return *reinterpret_cast<const f32x4*>(p.wpk + (size_t)cc * WCH + (piece < W_PIECES ? piece : 0) * 4);
  };
  auto store_w = [&](float* buf, int r, f32x4 v) {
    const int piece = r * THREADS + tid;
    if (piece < W_PIECES) *reinterpret_cast<f32x4*>(buf + piece * 4) = v;
  };

  // operand addresses: A = pixels (row 2*wave + mb of the tile, column l15), channels 16s + 4q .. +3
  //                    B = weights [tap][g = 4s + q][o = l15][4]
  const int x_lane = ((2 * wave) * kHalo + l15) * PSTR + 4 * q;
  const int w_lane = (q * NO + l15) * 4;

  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

  {
    f32x4 ir[IN_ROUNDS], wr[W_ROUNDS];
#pragma unroll
    for (int r = 0; r < IN_ROUNDS; ++r) ir[r] = load_in(r, 0);
#pragma unroll
    for (int r = 0; r < W_ROUNDS; ++r) wr[r] = load_w(r, 0);
#pragma unroll
    for (int r = 0; r < IN_ROUNDS; ++r) store_in(in_s, r, ir[r]);
#pragma unroll
    for (int r = 0; r < W_ROUNDS; ++r) store_w(w_s, r, wr[r]);
  }
  __syncthreads();

#pragma unroll 1
  for (int cc = 0; cc < NCC; ++cc) {
    const float* const ib = in_s + (cc & 1) * IN_FLOATS;
    const float* const wb = w_s + (cc & 1) * WCH;
    float* const ib_next = in_s + ((cc + 1) & 1) * IN_FLOATS;
    float* const wb_next = w_s + ((cc + 1) & 1) * WCH;
    const int ncc = cc + 1 < NCC ? cc + 1 : cc;       // last chunk re-fetches itself into the unused buffers
    f32x4 ir[IN_ROUNDS], wr[W_ROUNDS];
#pragma unroll
    for (int r = 0; r < IN_ROUNDS; ++r) ir[r] = load_in(r, ncc);
#pragma unroll
    for (int r = 0; r < W_ROUNDS; ++r) wr[r] = load_w(r, ncc);

#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
      for (int s = 0; s < KC / 16; ++s) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(wb + w_lane + ((tap * (KC / 4) + 4 * s) * NO) * 4);
        f32x4 a[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
          a[mb] = *reinterpret_cast<const f32x4*>(ib + x_lane + ((mb + dy) * kHalo + dx) * PSTR + 16 * s);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb][j], b[j], acc[mb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < IN_ROUNDS; ++r) store_in(ib_next, r, ir[r]);
#pragma unroll
    for (int r = 0; r < W_ROUNDS; ++r) store_w(wb_next, r, wr[r]);
    __syncthreads();
  }

  // epilogue: D[px][o]: lane owns channel o = l15 and pixels 4q .. 4q+3 of tile row 2*wave + mb
  const int o = l15;
  if (o < p.cout_real) {
    const float bias = p.bias[o];
    const int x = tx0 + 4 * q;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int y = ty0 + 2 * wave + mb;
      if (y < p.h && x < p.w) {
        const size_t idx = ((size_t)img * p.cout_real + o) * img_pix + (size_t)y * p.w + x;
        if (x + 4 <= p.w && (p.w & 3) == 0) {
          const f32x4 sk = *reinterpret_cast<const f32x4*>(p.aux + idx);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (acc[mb][e] + bias) + sk[e];
          *reinterpret_cast<f32x4*>(p.out + idx) = v;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (x + e < p.w) p.out[idx + e] = (acc[mb][e] + bias) + p.aux[idx + e];
        }
      }
    }
  }
}

template <int CIN>
static hipError_t launch_out_one(const ConvParams& p, hipStream_t stream) {
  auto kern = conv3x3_out_kernel<CIN>;
  static bool attr_set[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)outk::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  const long long tiles = (long long)p.n * p.tiles_x * p.tiles_y;
  if (tiles <= 0 || tiles > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(outk::THREADS), outk::LDS_BYTES, stream, p);
  return hipGetLastError();
}

hipError_t launch_conv3x3_out(const ConvParams& p, int feat, hipStream_t stream) {
  if (p.cout_real < 1 || p.cout_real > outk::NO) return hipErrorInvalidValue;
  if (feat == 128) return launch_out_one<128>(p, stream);
  if (feat == 256) return launch_out_one<256>(p, stream);
  return hipErrorInvalidValue;
}

}  // namespace dsen2
