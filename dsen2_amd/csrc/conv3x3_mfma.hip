// conv3x3_mfma.hip — im2col-free NHWC direct 3x3 convolution on the gfx950 fp32 matrix cores.
//
// Stands in for keras Conv2D(F, 3x3, padding='same') as the reference uses it
// (utils/DSen2Net.py:10,12,29,35) with the element-wise tail of each call site fused into the epilogue:
//   kEpiRelu      relu(conv + b)                      DSen2Net.py:10-11, :29
//   kEpiResidual  x + 0.1 * (conv + b)                DSen2Net.py:12-15
//   kEpiSkipNCHW  conv + b + low-res input, NCHW out  DSen2Net.py:35,38,41
//
// Mapping onto CDNA4 (one workgroup = 8 waves = 2 per SIMD, one workgroup per CU):
//   * GEMM view per workgroup:  D[o, px] = sum_k W[o, k] * X[k, px],  o = NT output channels (the A operand,
//     so that each lane ends up owning 4 CONSECUTIVE channels of one pixel -> 16-byte epilogue accesses),
//     px = a 16x16 pixel tile (the B operand), k = (tap, input channel).
//   * v_mfma_f32_32x32x2_f32 (exact f32, 64 cycles): one VGPR of A and one of B per instruction.  A lane
//     fetches 4 channels with ONE ds_read_b128 and feeds them to 4 successive MFMAs; because the k order of
//     a contraction is free, MFMA j of a step pairs channel 8s+j (lanes 0-31) with channel 8s+4+j (lanes 32-63)
//     on both operands.  No transposes, no im2col: a tap is an LDS address offset.
//   * LDS: the (16+2)^2 halo tile of KC input channels (pixel stride KC+4 floats keeps ds_read_b128
//     conflict-free across 16 consecutive pixels), double buffered across channel chunks, plus a
//     double-buffered KC x NT weight chunk per (tap, channel chunk) streamed from L2.  Zero padding of the
//     'same' convolution is materialised in LDS by predicated loads, never branched per MFMA.
//   * one s_barrier per (tap, chunk) step = per 64 MFMAs of each wave; next step's global loads are issued
//     before the MFMAs of the current one and land in LDS after them.
#include <string.h>

#include <vector>

#include "conv3x3_bf16_common.h"
#include "dsen2_internal.h"

namespace dsen2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CIN, int KC, int NT, int NWAVES>
struct ConvCfg {
  static constexpr int THREADS = 64 * NWAVES;
  static constexpr int NCC = CIN / KC;            // input-channel chunks
  static constexpr int NCHUNK = NCC * 9;          // (chunk, tap) steps
  static constexpr int PSTR = KC + 4;             // floats per halo pixel in LDS
  static constexpr int IN_FLOATS = kHaloPix * PSTR;
  static constexpr int NIBUF = NCC > 1 ? 2 : 1;
  static constexpr int WCH = KC * NT;             // floats per weight chunk
  static constexpr int QPP = KC / 4;              // 16-byte pieces per pixel
  static constexpr int IN_PIECES = kHaloPix * QPP;
  static constexpr int IN_ROUNDS = (IN_PIECES + THREADS - 1) / THREADS;
  static constexpr int W_PIECES = WCH / 4;
  static constexpr int W_ROUNDS = (W_PIECES + THREADS - 1) / THREADS;
  static constexpr int WN = NT >= 64 ? NT / 64 : 1;   // waves along output channels
  static constexpr int MB = NT / (32 * WN);           // 32-channel blocks per wave
  static constexpr int WP = NWAVES / WN;              // waves along pixels
  static constexpr int PB = 8 / WP;                   // 32-pixel (2 rows x 16) blocks per wave
  static constexpr size_t LDS_BYTES = (size_t)(NIBUF * IN_FLOATS + 2 * WCH) * sizeof(float);
  static_assert(CIN % KC == 0 && KC % 8 == 0, "channel chunking");
  static_assert(NWAVES % WN == 0 && 8 % WP == 0, "wave grid");
  static constexpr int RPT = (IN_ROUNDS + 8) / 9;     // input-prefetch rounds handled per tap
  static_assert((IN_FLOATS * 4) % 16 == 0 && (WCH * 4) % 16 == 0, "16-byte aligned LDS carve");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// CREAL (first layer only, CIN = KC = 16): number of real input channels (10 or 12; 0 = all CIN).  The padding
// channels are zeros in both operands; skipping their MFMAs leaves every accumulator's fma chain — and so every
// output bit — unchanged: channels 8.. are paired (8,9), (10,11) in one MFMA each, the same order in which the
// padded form adds them between its zero terms.
template <int CIN, int KC, int COUT, int NT, int EPI, int NWAVES, int WAVES_PER_SIMD, int CREAL = 0>
__global__ __launch_bounds__(64 * NWAVES, WAVES_PER_SIMD) void conv3x3_mfma_kernel(const ConvParams p) {
  using C = ConvCfg<CIN, KC, NT, NWAVES>;
  constexpr int kThreads = C::THREADS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const in_s = smem;                              // [NIBUF][324][PSTR]
  float* const w_s = smem + C::NIBUF * C::IN_FLOATS;     // [2][KC/4][NT][4]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave % C::WN;
  const int wp = wave / C::WN;
  const int l31 = lane & 31;
  const int hsel = lane >> 5;

  // workgroup -> tile.  Blocks b, b+8, b+16.. share an XCD (and its L2): give each XCD a contiguous run of
  // tiles so neighbouring tiles' halos and the weight stream hit in that L2.  Bijective for any grid size.
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int slab = blockIdx.y;
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const int img = lid / tiles_per_img;
  const int trem = lid - img * tiles_per_img;
  const int tyi = trem / p.tiles_x;
  const int ty0 = tyi * kTile;
  const int tx0 = (trem - tyi * p.tiles_x) * kTile;
  const size_t img_pix = (size_t)p.h * p.w;
  const float* const in_img = p.in + (size_t)img * img_pix * CIN;
  const float* const w_slab = p.wpk + (size_t)slab * C::NCHUNK * C::WCH;

  // ---- staging geometry (same for every channel chunk) ----
  int g_off[C::IN_ROUNDS];   // float offset of this thread's 16-byte piece inside the image, -1 = zero padding
  int s_off[C::IN_ROUNDS];   // float offset inside an LDS input buffer, -1 = no piece this round
#pragma unroll
  for (int r = 0; r < C::IN_ROUNDS; ++r) {
    const int piece = r * kThreads + tid;
    const int hp = piece / C::QPP, qq = piece - hp * C::QPP;
    const int hy = hp / kHalo, hx = hp - hy * kHalo;
    const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
    const bool have = piece < C::IN_PIECES;
    const bool inb = have && (unsigned)gy < (unsigned)p.h && (unsigned)gx < (unsigned)p.w;
    s_off[r] = have ? hp * C::PSTR + qq * 4 : -1;
    g_off[r] = inb ? (gy * p.w + gx) * CIN + qq * 4 : -1;
  }
  auto load_in = [&](int r, int cc) -> f32x4 {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (g_off[r] >= 0) v = *reinterpret_cast<const f32x4*>(in_img + g_off[r] + cc * KC);
    return v;
  };
  auto store_in = [&](float* buf, int r, f32x4 v) {
    if (s_off[r] >= 0) *reinterpret_cast<f32x4*>(buf + s_off[r]) = v;
  };
  auto load_w = [&](int chunk, f32x4 (&wr)[C::W_ROUNDS]) {
#pragma unroll
    for (int r = 0; r < C::W_ROUNDS; ++r) {
      const int piece = r * kThreads + tid;
      if (piece < C::W_PIECES) wr[r] = *reinterpret_cast<const f32x4*>(w_slab + (size_t)chunk * C::WCH + piece * 4);
    }
  };
  auto store_w = [&](float* buf, const f32x4 (&wr)[C::W_ROUNDS]) {
#pragma unroll
    for (int r = 0; r < C::W_ROUNDS; ++r) {
      const int piece = r * kThreads + tid;
      if (piece < C::W_PIECES) *reinterpret_cast<f32x4*>(buf + piece * 4) = wr[r];
    }
  };

  // ---- per-lane operand addresses ----
  // B (pixels): lane -> pixel (row l31>>4, col l31&15) of a 2x16 block; lanes 32-63 take channels +4.
  const int b_lane = ((l31 >> 4) * kHalo + (l31 & 15)) * C::PSTR + 4 * hsel + (2 * wp * C::PB) * kHalo * C::PSTR;
  // A (weights): [g = 2s + hsel][o][4]
  const int a_lane = (hsel * NT + wn * (32 * C::MB) + l31) * 4;

  f32x16 acc[C::MB][C::PB];
#pragma unroll
  for (int mb = 0; mb < C::MB; ++mb)
#pragma unroll
    for (int pb = 0; pb < C::PB; ++pb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mb][pb][e] = 0.f;

  // ---- prologue: input chunk 0 and weight chunk 0 ----
  {
    f32x4 ir[C::IN_ROUNDS];
#pragma unroll
    for (int r = 0; r < C::IN_ROUNDS; ++r) ir[r] = load_in(r, 0);
    f32x4 wr[C::W_ROUNDS];
    load_w(0, wr);
#pragma unroll
    for (int r = 0; r < C::IN_ROUNDS; ++r) store_in(in_s, r, ir[r]);
    store_w(w_s, wr);
  }
  __syncthreads();

#pragma unroll 1
  for (int cc = 0; cc < C::NCC; ++cc) {
    const float* const ib = in_s + (C::NIBUF > 1 ? (cc & 1) * C::IN_FLOATS : 0);
    float* const ib_next = in_s + (C::NIBUF > 1 ? ((cc + 1) & 1) * C::IN_FLOATS : 0);
    const bool more_in = (C::NCC > 1) && (cc + 1 < C::NCC);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int chunk = cc * 9 + tap;
      const float* const wb = w_s + (chunk & 1) * C::WCH;
      float* const wb_next = w_s + ((chunk + 1) & 1) * C::WCH;
      const bool more_w = chunk + 1 < C::NCHUNK;

      // issue next step's global loads first: their latency hides behind this step's MFMAs
      f32x4 wr[C::W_ROUNDS];
      if (more_w) load_w(chunk + 1, wr);
      f32x4 ir[C::RPT];
#pragma unroll
      for (int q = 0; q < C::RPT; ++q) {
        const int r = tap * C::RPT + q;
        if (r < C::IN_ROUNDS && more_in) ir[q] = load_in(r < C::IN_ROUNDS ? r : 0, cc + 1);
      }

      const int dy = tap / 3, dx = tap - dy * 3;
      const float* const bp = ib + b_lane + (dy * kHalo + dx) * C::PSTR;
      const float* const ap = wb + a_lane;
#pragma unroll
      for (int s = 0; s < KC / 8; ++s) {
        if constexpr (CREAL > 0) {
          static_assert(CREAL % 2 == 0 && CREAL > 8 && CREAL <= 16 && KC == 16 && CIN == 16, "first-layer form");
          if (s == 1) {
            // channels 8 .. CREAL-1: lanes 0-31 supply channel 8 + 2m, lanes 32-63 channel 9 + 2m (both from k-group 2)
#pragma unroll
            for (int m = 0; m < (CREAL - 8) / 2; ++m) {
              float a1[C::MB], b1[C::PB];
#pragma unroll
              for (int mb = 0; mb < C::MB; ++mb) a1[mb] = ap[(2 * NT + mb * 32) * 4 - hsel * NT * 4 + 2 * m + hsel];
#pragma unroll
              for (int pb = 0; pb < C::PB; ++pb) b1[pb] = bp[pb * 2 * kHalo * C::PSTR + 8 - 4 * hsel + 2 * m + hsel];
#pragma unroll
              for (int mb = 0; mb < C::MB; ++mb)
#pragma unroll
                for (int pb = 0; pb < C::PB; ++pb)
                  acc[mb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[mb], b1[pb], acc[mb][pb], 0, 0, 0);
            }
            continue;
          }
        }
        f32x4 a[C::MB], b[C::PB];
#pragma unroll
        for (int mb = 0; mb < C::MB; ++mb) a[mb] = *reinterpret_cast<const f32x4*>(ap + (2 * s * NT + mb * 32) * 4);
#pragma unroll
        for (int pb = 0; pb < C::PB; ++pb)
          b[pb] = *reinterpret_cast<const f32x4*>(bp + pb * 2 * kHalo * C::PSTR + 8 * s);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mb = 0; mb < C::MB; ++mb)
#pragma unroll
            for (int pb = 0; pb < C::PB; ++pb)
              acc[mb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb][j], b[pb][j], acc[mb][pb], 0, 0, 0);
      }

      if (more_w) store_w(wb_next, wr);
#pragma unroll
      for (int q = 0; q < C::RPT; ++q) {
        const int r = tap * C::RPT + q;
        if (r < C::IN_ROUNDS && more_in) store_in(ib_next, r < C::IN_ROUNDS ? r : 0, ir[q]);
      }
      __syncthreads();
    }
  }

  // ---- epilogue ----
  // 32x32 D layout: lane owns pixel column l31 and rows (reg&3) + 8*(reg>>2) + 4*hsel, i.e. register quad g
  // holds output channels 8g + 4*hsel .. +3 of this pixel: one 16-byte access per quad.
#pragma unroll
  for (int pb = 0; pb < C::PB; ++pb) {
    const int blk = wp * C::PB + pb;
    const int y = ty0 + 2 * blk + (l31 >> 4);
    const int x = tx0 + (l31 & 15);
    if (y < p.h && x < p.w) {
      const size_t pix = (size_t)img * img_pix + (size_t)y * p.w + x;
#pragma unroll
      for (int mb = 0; mb < C::MB; ++mb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = slab * NT + wn * (32 * C::MB) + mb * 32 + 8 * g + 4 * hsel;
          f32x4 v = {acc[mb][pb][4 * g], acc[mb][pb][4 * g + 1], acc[mb][pb][4 * g + 2], acc[mb][pb][4 * g + 3]};
          v += *reinterpret_cast<const f32x4*>(p.bias + c0);
          if constexpr (EPI == kEpiRelu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            *reinterpret_cast<f32x4*>(p.out + pix * COUT + c0) = v;
          } else if constexpr (EPI == kEpiReluSplit) {
            // the same values as blocked (hi, lo) planes [n][C/8][h][w][8] (conv3x3_body16w.hip): this lane's 4 channels
            // are bytes 8*hsel .. 8*hsel+7 of the pixel's 16-byte piece in block c0 >> 3; lanes l and l + 32 complete it
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            const float f0 = v[0], f1 = v[1], f2 = v[2], f3 = v[3];
            unsigned h01, l01, h23, l23;
            bf16k::split2(__float_as_uint(f0), __float_as_uint(f1), h01, l01);
            bf16k::split2(__float_as_uint(f2), __float_as_uint(f3), h23, l23);
            const size_t off = (((size_t)img * (COUT / 8) + (c0 >> 3)) * img_pix + (size_t)y * p.w + x) * 16 + (c0 & 7) * 2;
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(p.out) + off) = u32x2{h01, h23};
            *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(p.out2) + off) = u32x2{l01, l23};
          } else if constexpr (EPI == kEpiResidual) {
            const f32x4 res = *reinterpret_cast<const f32x4*>(p.aux + pix * COUT + c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __fadd_rn(res[e], __fmul_rn(v[e], p.res_scale));
            *reinterpret_cast<f32x4*>(p.out + pix * COUT + c0) = v;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int oc = c0 + e;
              if (oc < p.cout_real) {
                const size_t idx = ((size_t)img * p.cout_real + oc) * img_pix + (size_t)y * p.w + x;
                p.out[idx] = v[e] + p.aux[idx];
              }
            }
          }
        }
      }
    }
  }
}

template <int CIN, int KC, int COUT, int NT, int EPI, int NWAVES = 8, int WAVES_PER_SIMD = 2, int CREAL = 0>
static hipError_t launch_one(const ConvParams& p, hipStream_t stream) {
  using C = ConvCfg<CIN, KC, NT, NWAVES>;
  auto kern = conv3x3_mfma_kernel<CIN, KC, COUT, NT, EPI, NWAVES, WAVES_PER_SIMD, CREAL>;
  static KernelOnce once;
  hipError_t e = once.prepare(reinterpret_cast<const void*>(kern), C::LDS_BYTES, nullptr);
  if (e != hipSuccess) return e;
  const long long tiles = (long long)p.n * p.tiles_x * p.tiles_y;
  if (tiles <= 0 || tiles > 0x7fffffffLL) return hipErrorInvalidValue;
  dim3 grid((unsigned)tiles, COUT / NT, 1);
  hipLaunchKernelGGL(kern, grid, dim3(C::THREADS), C::LDS_BYTES, stream, p);
  return hipGetLastError();
}

bool conv_pack_geometry(int cin, int cout, int epilogue, const Tuning& tune, PackGeom* g) {
  if (cin <= 0 || cout <= 0) return false;
  if (epilogue == kEpiSkipNCHW) {
    if (cout > 32 || (cin != 128 && cin != 256)) return false;
    // variant 8: conv3x3_out_mfma.hip where the shape fits, else conv3x3_out.hip (the buffer holds both packings);
    // variant 9: conv3x3_out.hip always
    if (cout <= 8 && (tune.out_variant == 2 || tune.out_variant == 3)) {
      *g = PackGeom{16, 8, cin, 8, tune.out_variant == 2 ? 8 : 9};
      return true;
    }
    *g = PackGeom{32, 32, cin, 32, 0};
    return true;
  }
  if (cout != 128 && cout != 256) return false;
  if (cin <= 16) {
    // first layer: variant = number of real channels whose MFMAs are issued (10 / 12: the Sentinel-2 band groups;
    // 0 = all 16 padded channels, the reference structure)
    *g = PackGeom{16, 128, 16, cout, (tune.body_variant != 0 && (cin == 10 || cin == 12)) ? cin : 0};
    return true;
  }
  if (cin == cout) {
    // every structure of the F->F body convolution reads the same packing (KC = 32, NT = 128):
    // 11-14 = conv3x3_body32.hip sub-variants 0-3 (14 = default); 0 = one tile per workgroup (this file)
    *g = PackGeom{32, 128, cin, cout, tune.body_variant >= 11 && tune.body_variant <= 14 ? tune.body_variant : 0};
    return true;
  }
  return false;
}

size_t packed_weight_floats(const PackGeom& g) {
  return (size_t)9 * g.cin_pad * g.cout_pad + (g.variant == 8 ? out_mfma_weight_floats(g.cin_pad) : 0);
}

void pack_conv_weights_host(const float* k, int cin, int cout, const PackGeom& g, float* dst) {
  if (g.variant == 8 || g.variant == 9) {          // the output kernels have their own operand orders
    pack_out_valu_weights_host(k, cin, cout, dst);
    if (g.variant == 8) pack_out_mfma_weights_host(k, cin, cout, dst + (size_t)9 * cin * 8);
    return;
  }
  const int ncc = g.cin_pad / g.kc, nslab = g.cout_pad / g.nt, ng = g.kc / 4;
  size_t i = 0;
  for (int slab = 0; slab < nslab; ++slab)
    for (int cc = 0; cc < ncc; ++cc)
      for (int tap = 0; tap < 9; ++tap)
        for (int gg = 0; gg < ng; ++gg)
          for (int o = 0; o < g.nt; ++o)
            for (int j = 0; j < 4; ++j, ++i) {
              const int c = cc * g.kc + 4 * gg + j, oc = slab * g.nt + o;
              dst[i] = (c < cin && oc < cout) ? k[((size_t)tap * cin + c) * cout + oc] : 0.f;
            }
}

static inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

void pack_conv_weights_bf16_host(const float* k, int cin, int cout, int chunk_ch, bool perm16, uint16_t* dst) {
  // [slab][cc (chunk_ch channels)][step][g (8 channels)][o (128)][j (8)]: one (slab, cc, step) chunk is the LDS image.
  // perm16 (conv3x3_body16w.hip's format): step s carries tap (dy, dx) = (s % 3, s / 3) — the kernel walks the taps
  // dx-major to keep its pixel-row fragments over the three dy of one dx; otherwise step = tap.
  const int ncc = cin / chunk_ch, nslab = cout / 128, ng = chunk_ch / 8;
  size_t i = 0;
  for (int slab = 0; slab < nslab; ++slab)
    for (int cc = 0; cc < ncc; ++cc)
      for (int step = 0; step < 9; ++step) {
        const int tap = perm16 ? (step % 3) * 3 + step / 3 : step;
        for (int g = 0; g < ng; ++g)
          for (int o = 0; o < 128; ++o)
            for (int j = 0; j < 8; ++j, ++i) {
              const int ch = perm16 ? 32 * (o >> 5) + 8 * ((o & 15) >> 2) + 4 * ((o >> 4) & 1) + (o & 3) : o;
              const int c = cc * chunk_ch + 8 * g + j, oc = slab * 128 + ch;
              dst[i] = f32_to_bf16_rne(k[((size_t)tap * cin + c) * cout + oc]);
            }
      }
}

void pack_conv_weights_bf16x3_host(const float* k, int cin, int cout, uint16_t* dst) {
  // a virtual (3, 3, 3*cin, cout) kernel: input chunk 3*cc + j of 32 channels = plane (wh, wl, wh)[j] of real chunk cc — the
  // order in which conv3x3_body16w.hip (X3) walks the activation planes (xh, xh, xl)
  const int vcin = 3 * cin;
  std::vector<float> v((size_t)9 * vcin * cout);
  for (int tap = 0; tap < 9; ++tap)
    for (int c = 0; c < cin; ++c)
      for (int o = 0; o < cout; ++o) {
        const float w = k[((size_t)tap * cin + c) * cout + o];
        const uint16_t hb = f32_to_bf16_rne(w);
        uint32_t hu = (uint32_t)hb << 16;
        float wh;
        memcpy(&wh, &hu, 4);
        const float wl = w - wh;                       // exact; rounded to bf16 by the packer below
        const int cc = c / 32, j = c % 32;
        float* base = v.data() + ((size_t)tap * vcin + (size_t)cc * 96 + j) * cout + o;
        base[0] = wh;
        base[(size_t)32 * cout] = wl;
        base[(size_t)64 * cout] = wh;
      }
  pack_conv_weights_bf16_host(v.data(), vcin, cout, 32, true, dst);
}

hipError_t launch_conv3x3(const ConvParams& p, const PackGeom& geom, int epilogue, int ablate, hipStream_t stream) {
  const int cin_pad = geom.cin_pad, cout_pad = geom.cout_pad;
  if ((geom.variant == 8 || geom.variant == 9) && epilogue == kEpiSkipNCHW) {
    if (geom.variant == 8) {
      ConvParams pm = p;
      pm.wpk = p.wpk + (size_t)9 * cin_pad * 8;
      bool taken = false;
      const hipError_t e = launch_conv3x3_out_mfma(pm, cin_pad, stream, &taken, ablate);
      if (e != hipSuccess || taken) return e;
    }
    return launch_conv3x3_out_valu(p, cin_pad, stream);
  }
  // 11-14: the DMA-fed kernel (conv3x3_body32.hip) and its sub-variants.  An image it cannot address (>= 2 GiB of
  // activations) is an error, not a silent switch of kernels: dsen2's entry points reject such shapes up front.
  if (geom.variant >= 11 && geom.variant <= 14 && cin_pad == cout_pad && epilogue != kEpiSkipNCHW)
    return launch_conv3x3_body32(p, cin_pad, epilogue, geom.variant - 11, ablate, stream);
  if (ablate != 0) return hipErrorInvalidValue;
  if (cin_pad == 16 && epilogue == kEpiReluSplit) {          // first convolution of a precision-1 model
    if (!p.out2) return hipErrorInvalidValue;
#define DSEN2_FIRST(CO, CR) return launch_one<16, 16, CO, 128, kEpiReluSplit, 8, 2, CR>(p, stream);
    if (cout_pad == 128) { if (geom.variant == 10) DSEN2_FIRST(128, 10) if (geom.variant == 12) DSEN2_FIRST(128, 12) DSEN2_FIRST(128, 0) }
    if (cout_pad == 256) { if (geom.variant == 10) DSEN2_FIRST(256, 10) if (geom.variant == 12) DSEN2_FIRST(256, 12) DSEN2_FIRST(256, 0) }
#undef DSEN2_FIRST
    return hipErrorInvalidValue;
  }
  if (cin_pad == 16 && epilogue == kEpiRelu && (geom.variant == 10 || geom.variant == 12)) {
    if (cout_pad == 128)
      return geom.variant == 10 ? launch_one<16, 16, 128, 128, kEpiRelu, 8, 2, 10>(p, stream)
                                : launch_one<16, 16, 128, 128, kEpiRelu, 8, 2, 12>(p, stream);
    if (cout_pad == 256)
      return geom.variant == 10 ? launch_one<16, 16, 256, 128, kEpiRelu, 8, 2, 10>(p, stream)
                                : launch_one<16, 16, 256, 128, kEpiRelu, 8, 2, 12>(p, stream);
  }
#define DSEN2_CASE(CI, KC_, CO, NT_, EP) \
  if (cin_pad == CI && cout_pad == CO && epilogue == EP) return launch_one<CI, KC_, CO, NT_, EP>(p, stream);
  DSEN2_CASE(16, 16, 128, 128, kEpiRelu)
  DSEN2_CASE(128, 32, 128, 128, kEpiRelu)
  DSEN2_CASE(128, 32, 128, 128, kEpiResidual)
  DSEN2_CASE(128, 32, 32, 32, kEpiSkipNCHW)
  DSEN2_CASE(16, 16, 256, 128, kEpiRelu)
  DSEN2_CASE(256, 32, 256, 128, kEpiRelu)
  DSEN2_CASE(256, 32, 256, 128, kEpiResidual)
  DSEN2_CASE(256, 32, 32, 32, kEpiSkipNCHW)
#undef DSEN2_CASE
  return hipErrorInvalidValue;
}

}  // namespace dsen2
