// conv3x3_out.hip — the network's last convolution: F -> Cout (6 or 2 channels) 3x3 'same' + bias + the
// low-resolution skip input, NHWC in, NCHW out (utils/DSen2Net.py:35,38,41) — on the VECTOR units.
//
// With 6 (or 2) output channels a matrix-core tile is mostly padding: a 32-wide MFMA block 81 % (94 %), the
// 16x16x4 form (round 1's kernel: 171 us at the bench config) still 62 %.  A vector kernel pads nothing (v_fma_f32
// with the weight as scalar operand: 78.6 TFLOP/s, half the matrix rate): 122 us, same bits.  The packed
// v_pk_fma_f32 form and other restructurings are in experiments/README.md (bit-identical, none faster).
#include "dsen2_internal.h"

namespace dsen2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Cout <= 8 (more output channels fall back to the padded 32-wide MFMA block of conv3x3_mfma.hip).  One thread = one pixel x
// half of the output channels (wave parity h: outputs h, h+2, h+4, ..): NSLOT accumulators, v_fma_f32 with the
// weight as the instruction's scalar operand.  Weights are re-packed [chunk][tap][h][16 chain positions][4 slots]
// so that one s_load_dwordx4 per chain position feeds NSLOT FMAs; the chain order (channel 4q + j at position
// 4j + q of a 16-channel chunk) is the k order of round 1's v_mfma_f32_16x16x4_f32 kernel (an f32 MFMA is an fma
// chain over its k), so the results are the same bits (checked against the round-1 library on whole networks).
// Input: 18x18 halo tile of 16 channels, double buffered through registers; 51.8 KB of LDS, two workgroups per CU.
namespace outv {
constexpr int KC = 16;
constexpr int THREADS = 512;
constexpr int PSTR = KC + 4;
constexpr int IN_FLOATS = kHaloPix * PSTR;
constexpr int IN_PIECES = kHaloPix * (KC / 4);
constexpr int IN_ROUNDS = (IN_PIECES + THREADS - 1) / THREADS;
constexpr size_t LDS_BYTES = (size_t)(2 * IN_FLOATS) * sizeof(float);     // 51,840 B
constexpr int WCHUNK = 9 * 2 * 16 * 4;                                    // floats per 16-channel chunk of weights
}  // namespace outv

template <int CIN, int NSLOT>
__global__ __launch_bounds__(outv::THREADS, 2) void conv3x3_out_valu_kernel(const ConvParams p) {
  using namespace outv;
  constexpr int NCC = CIN / KC;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const in_s = smem;                    // [2][324][PSTR]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = wave & 1;                      // which half of the output channels
  // A wave's 64 pixels are rows {2w, 2w + 8, 2w + 1, 2w + 9} x 16 columns (lane groups of 16): a ds_read_b128 is served in
  // the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, {32-35, 44-47, 52-59}, {36-43, 48-51, 60-63}
  // (MI355X_MICROARCH.md §LDS), each mixing two 16-lane rows; rows 8 apart are 8 x 18 x 5 = 720 = 0 mod 16 slots apart, so
  // a group's 16 lanes hit 16 different 16-byte slots (5 slots per pixel).  With four ADJACENT rows per wave (90 slots
  // apart) every group was 2-way conflicted: SQ_LDS_BANK_CONFLICT 50 % of this kernel's LDS cycles.
  const int row = 2 * (wave >> 1) + 8 * ((lane >> 4) & 1) + (lane >> 5), col = lane & 15;

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

  const int x_lane = (row * kHalo + col) * PSTR;
  const float* const wv = p.wpk + h * 64;        // [chunk][tap][h][16][4]: wave-uniform -> scalar loads

  float acc[NSLOT];
#pragma unroll
  for (int m = 0; m < NSLOT; ++m) acc[m] = 0.f;

  {
    f32x4 ir[IN_ROUNDS];
#pragma unroll
    for (int r = 0; r < IN_ROUNDS; ++r) ir[r] = load_in(r, 0);
#pragma unroll
    for (int r = 0; r < IN_ROUNDS; ++r) store_in(in_s, r, ir[r]);
  }
  __syncthreads();

#pragma unroll 1
  for (int cc = 0; cc < NCC; ++cc) {
    const float* const ib = in_s + (cc & 1) * IN_FLOATS;
    float* const ib_next = in_s + ((cc + 1) & 1) * IN_FLOATS;
    const int ncc = cc + 1 < NCC ? cc + 1 : cc;       // last chunk re-fetches itself into the unused buffer
    f32x4 ir[IN_ROUNDS];
#pragma unroll
    for (int r = 0; r < IN_ROUNDS; ++r) ir[r] = load_in(r, ncc);
    const float* const wc = wv + (size_t)cc * WCHUNK;
    // 18 half taps of 8 chain positions: the 32 weights of half tap k+1 are fetched (two s_load_dwordx16) while half
    // tap k's FMAs run.  The scheduling barriers keep hipcc from hoisting a whole chunk's 576 scalars at once, which
    // it then has to park in VGPR lanes (v_writelane / v_readlane per weight).
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    // constant address space + wave-uniform address = scalar loads (s_load_dwordx16) into SGPRs, whatever else the
    // optimiser believes about aliasing
    typedef const __attribute__((address_space(4))) f32x16* const_f32x16_ptr;
    auto load_half = [&](int k, f32x16& a, f32x16& b) __attribute__((always_inline)) {
      const float* const wh = wc + (k >> 1) * 128 + (k & 1) * 32;
      a = *reinterpret_cast<const_f32x16_ptr>(reinterpret_cast<size_t>(wh));
      b = *reinterpret_cast<const_f32x16_ptr>(reinterpret_cast<size_t>(wh + 16));
    };
    f32x16 wa[2], wb[2];
    load_half(0, wa[0], wb[0]);
    f32x4 x[4];
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      const int tap = k >> 1;
      if ((k & 1) == 0) {
        const int dy = tap / 3, dx = tap - dy * 3;
        const float* const xp = ib + x_lane + (dy * kHalo + dx) * PSTR;
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const f32x4*>(xp + 4 * q);
      }
      // (waiting for this half tap's operands BEFORE issuing the next fetches — so that they fly under the FMAs —
      // measured slower: 139 vs 122 us; four waves per SIMD already cover the scalar-load latency)
      if (k + 1 < 18) load_half(k + 1, wa[(k + 1) & 1], wb[(k + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ii = 0; ii < 8; ++ii) {             // chain position i = 4j + q  <->  channel 4q + j
        const int i = 8 * (k & 1) + ii;
        const int q = i & 3, j = i >> 2;
        const f32x16& wsrc = ii < 4 ? wa[k & 1] : wb[k & 1];
#pragma unroll
        for (int m = 0; m < NSLOT; ++m) acc[m] = __builtin_fmaf(x[q][j], wsrc[4 * (ii & 3) + m], acc[m]);
      }
      // pin the half tap's FMAs here (without this the optimiser sinks every FMA below all scalar loads)
#pragma unroll
      for (int m = 0; m < NSLOT; ++m) asm volatile("" : "+v"(acc[m]));
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < IN_ROUNDS; ++r) store_in(ib_next, r, ir[r]);
    __syncthreads();
  }

  const int y = ty0 + row, xg = tx0 + col;
  if (y < p.h && xg < p.w) {
#pragma unroll
    for (int m = 0; m < NSLOT; ++m) {
      const int o = 2 * m + h;
      if (o < p.cout_real) {
        const size_t idx = ((size_t)img * p.cout_real + o) * img_pix + (size_t)y * p.w + xg;
        p.out[idx] = (acc[m] + p.bias[o]) + p.aux[idx];
      }
    }
  }
}

template <int CIN, int NSLOT>
static hipError_t launch_out_valu_one(const ConvParams& p, hipStream_t stream) {
  auto kern = conv3x3_out_valu_kernel<CIN, NSLOT>;
  static KernelOnce once;
  hipError_t e = once.prepare(reinterpret_cast<const void*>(kern), outv::LDS_BYTES, nullptr);
  if (e != hipSuccess) return e;
  const long long tiles = (long long)p.n * p.tiles_x * p.tiles_y;
  if (tiles <= 0 || tiles > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(outv::THREADS), outv::LDS_BYTES, stream, p);
  return hipGetLastError();
}

// weights packed by pack_out_valu_weights_host (PackGeom variant 8)
hipError_t launch_conv3x3_out_valu(const ConvParams& p, int feat, hipStream_t stream) {
  if (p.cout_real < 1 || p.cout_real > 8) return hipErrorInvalidValue;
  const int nslot = (p.cout_real + 1) / 2;
#define DSEN2_OUTV(F, S) if (feat == F && nslot == S) return launch_out_valu_one<F, S>(p, stream);
  DSEN2_OUTV(128, 1) DSEN2_OUTV(128, 2) DSEN2_OUTV(128, 3) DSEN2_OUTV(128, 4)
  DSEN2_OUTV(256, 1) DSEN2_OUTV(256, 2) DSEN2_OUTV(256, 3) DSEN2_OUTV(256, 4)
#undef DSEN2_OUTV
  return hipErrorInvalidValue;
}

// kernel HWIO (3,3,cin,cout<=8) -> [chunk of 16 ch][tap][h][chain position i = 4j + q][slot m]:
//   value = K[tap][16*chunk + 4q + j][2m + h]   (0 where 2m + h >= cout);  9*cin*8 floats
void pack_out_valu_weights_host(const float* k, int cin, int cout, float* dst) {
  size_t n = 0;
  for (int cc = 0; cc < cin / 16; ++cc)
    for (int tap = 0; tap < 9; ++tap)
      for (int h = 0; h < 2; ++h)
        for (int i = 0; i < 16; ++i)
          for (int m = 0; m < 4; ++m, ++n) {
            const int c = 16 * cc + 4 * (i & 3) + (i >> 2), o = 2 * m + h;
            dst[n] = o < cout ? k[((size_t)tap * cin + c) * cout + o] : 0.f;
          }
}

}  // namespace dsen2
