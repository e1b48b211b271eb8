// conv3x3_first.hip — the network's FIRST convolution (utils/DSen2Net.py:24-29): Concatenate(axis=1) of the two or
// three NCHW inputs, Conv2D(F, 3x3, 'same') + bias + ReLU -> NHWC fp32, exact fp32 arithmetic (precision 0).  The bf16-operand
// modes run this layer on the bf16 matrix cores and write the residual stream's planes: conv3x3_first16.hip.
//
// Round 2 ran this layer as pack_inputs_kernel (NCHW -> NHWC16, a 54 MB round trip) + conv3x3_mfma_kernel<16, 16, ...>:
// one 16x16-pixel tile per workgroup, the 72 KB of weights streamed from L2 for EVERY tile with a barrier per tap.
// With K = 9 x 10 the layer has 77 us of MFMA work and 268 MB of output at the bench config: it is bound by how well
// the two overlap, not by either.  This kernel:
//   * is persistent (one workgroup of 8 waves per CU walks tiles lid, lid + G, ...): the weights of the workgroup's output
//     slab (the three k-groups the 10 / 12 real channels use: 54 KB) and the bias are copied to LDS ONCE, the tap loop
//     has no barrier — one barrier per tile;
//   * reads the NCHW inputs directly: the halo tile of the next item is gathered plane by plane (18-pixel row
//     segments, coalesced) into registers while the current tile computes, then written to the other half of a double
//     buffer in the MFMA's operand layout ([halo pixel][channel], 20-float pixel pitch) — the Concatenate is an address
//     computation;
//   * DEFERS its epilogue like the body kernel: a finished tile's accumulators are copied to a second register set and
//     written out two 16-byte pieces per tap inside the NEXT tile's tap loop, so the 128 KB of stores per tile leave
//     under MFMAs instead of in a phase of their own (measured before: MFMA loop alone 97 us, stores alone 47 us, the
//     two in sequence 126 us — profiles/archive/r03_d_first_conv_ablation.log).
// The MFMA sequence per accumulator is exactly conv3x3_mfma_kernel's first-layer form (per tap: channels (j, 4 + j) for
// j = 0..3, then the pairs (8, 9)(, (10, 11)); taps in order), so every output bit is the same
// (tests/test_gpu_forward.py::test_first_layer_without_padding_mfmas_gives_the_same_bits pins it against the generic kernel).
#include "conv3x3_bf16_common.h"
#include "dsen2_internal.h"

namespace dsen2 {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace first {
constexpr int THREADS = 512;                        // 8 waves: 2 (64-channel halves) x 4 (pixel quarters: 4 rows x 16)
constexpr int NT = 128;                             // output channels per item (slab)
constexpr int PSTR = 20;                            // floats per halo pixel in LDS: conflict-free ds_read_b128 over 16 pixels
constexpr int IN_FLOATS = kHaloPix * PSTR;          // 6480
constexpr int WCH = 16 * NT;                        // floats per tap of packed weights [g: 4][o: 128][j: 4]
constexpr int WCH3 = 12 * NT;                       // ... of which k-groups 0-2 (channels 0-11) are kept in LDS
constexpr int W_FLOATS = 9 * WCH;                   // one slab, all taps, in global memory
constexpr int W3_FLOATS = 9 * WCH3;                 // 13,824 floats = 54 KB in LDS
constexpr size_t LDS_BYTES = (size_t)(W3_FLOATS + 2 * IN_FLOATS + NT) * sizeof(float);      // 107,648 B
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
}  // namespace first

}  // namespace

// CREAL: real input channels (10 = 4 + 6 or 12 = 4 + 6 + 2; c10 + c20 + c60 must equal it).  EPI: kEpiRelu (p.out fp32
// NHWC).  p.in = x10, p.aux = x20, p.diag unused; x60 and the channel counts come in `f`.
// ABL (diagnostic builds, timing only): 1 no stores, 2 no MFMAs, 4 no input gather.
template <int CREAL, int COUT, int EPI, int ABL = 0>
__global__ __launch_bounds__(first::THREADS, 2) void conv3x3_first_kernel(const ConvParams p, const FirstInputs f, const int n_items) {
  using namespace first;
  constexpr int NS = COUT / NT;
  constexpr int MB = 2, PB = 2;
  static_assert(CREAL % 2 == 0 && CREAL > 8 && CREAL <= 16, "first-layer form");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const w_s = smem;                                          // [9][3][128][4]
  float* const in_s = smem + W3_FLOATS;                             // [2][324][PSTR]
  float* const bias_s = in_s + 2 * IN_FLOATS;                       // [128]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1;
  const int wp = wave >> 1;
  const int l31 = lane & 31;
  const int hsel = lane >> 5;

  // persistent schedule (XCD-contiguous like the body kernels); a workgroup keeps ONE output slab: item = tile * NS + slab
  const int G = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  if (lid >= n_items) return;
  const int slab = lid % NS;                                        // G is a multiple of NS (launcher)
  const int my_items = (n_items - lid + G - 1) / G;
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const size_t plane = (size_t)p.h * p.w;

  // ---- weights of this slab: global -> LDS once (the packed layout is the LDS image) ----
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(p.wpk + (size_t)slab * W_FLOATS);
    f32x4* dst = reinterpret_cast<f32x4*>(w_s);
    // per tap 3 x 128 pieces of 16 bytes (k-groups 0-2 of the packed [g: 4][o: 128][4])
    for (int i = tid; i < 9 * (WCH3 / 4); i += THREADS) {
      const int tap = i / (WCH3 / 4), q = i - tap * (WCH3 / 4);
      dst[i] = src[tap * (WCH / 4) + q];
    }
    if (tid < NT) bias_s[tid] = p.bias[slab * NT + tid];
  }

  // ---- gather geometry of one halo tile (the same for every tile) ----
  // Per input tensor T (10 m: 4 channels, 20 m: 6, 60 m: CREAL - 10) the CT x 324 values of a halo tile are fetched in
  // rounds of 512 threads, channel outer / halo pixel inner: consecutive lanes read consecutive pixels of an 18-pixel row
  // segment of ONE plane, through a per-image buffer descriptor (an out-of-range offset returns the zero padding).
  // One register per round: pk = hx | hy << 5 | (channel inside its tensor) << 10 | (LDS float offset) << 13, negative = no element.
  constexpr int C10 = 4, C20 = 6, C60 = CREAL - 10;
  constexpr int R10 = (C10 * kHaloPix + THREADS - 1) / THREADS, R20 = (C20 * kHaloPix + THREADS - 1) / THREADS,
                R60 = (C60 * kHaloPix + THREADS - 1) / THREADS;
  constexpr int ROUNDS = R10 + R20 + R60;                           // 7 (10 channels) or 9 (12)
  int pk[ROUNDS];
  {
    auto setup = [&](int r0, int rounds, int ct, int cbase) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < rounds; ++r) {
        const int e = r * THREADS + tid;
        const int c = e / kHaloPix, hp = e - c * kHaloPix;
        const int hy = hp / kHalo, hx = hp - hy * kHalo;
        const bool have = e < ct * kHaloPix;
        pk[r0 + r] = have ? hx | hy << 5 | c << 10 | (hp * PSTR + cbase + c) << 13 : -1;
      }
    };
    setup(0, R10, C10, 0);
    setup(R10, R20, C20, C10);
    if constexpr (R60 > 0) setup(R10 + R20, R60, C60, C10 + C20);
  }
  struct Tile { int img, ty0, tx0; };
  auto tile_of = [&](int item) -> Tile {
    const int tile = item / NS;
    const int img = tile / tiles_per_img;
    const int trem = tile - img * tiles_per_img;
    const int tyi = trem / p.tiles_x;
    return Tile{img, tyi * kTile, (trem - tyi * p.tiles_x) * kTile};
  };
  auto gather = [&](const Tile& t, float (&v)[ROUNDS]) __attribute__((always_inline)) {
    auto fetch = [&](int r0, int rounds, const float* x, int ct) __attribute__((always_inline)) {
      const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) + (size_t)t.img * ct * plane, 0,
                                                          (unsigned)(ct * plane * 4), 0x00020000);
#pragma unroll
      for (int r = 0; r < rounds; ++r) {
        int k = pk[r0 + r];
        asm volatile("" : "+v"(k));      // derive the addresses here, every tile: hoisted out of the item loop they are spilled
        const int gy = t.ty0 - 1 + ((k >> 5) & 31), gx = t.tx0 - 1 + (k & 31);
        const bool inb = k >= 0 && (unsigned)gy < (unsigned)p.h && (unsigned)gx < (unsigned)p.w;
        const unsigned voff = inb ? (unsigned)((((k >> 10) & 7) * (int)plane + gy * p.w + gx) * 4) : 0x80000000u;
        v[r0 + r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, 0, 0));
      }
    };
    fetch(0, R10, p.in, C10);
    fetch(R10, R20, p.aux, C20);
    if constexpr (R60 > 0) fetch(R10 + R20, R60, f.x60, C60);
  };
  auto scatter = [&](float* buf, const float (&v)[ROUNDS]) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r)
      if (pk[r] >= 0) buf[pk[r] >> 13] = v[r];
  };

  // ---- per-lane operand addresses (floats): conv3x3_mfma_kernel<16, 16, ., 128, ., 8, 2, CREAL>'s ----
  // B (pixels): lane -> pixel (row l31 >> 4, column l31 & 15) of a 2 x 16 block; lanes 32-63 take channels + 4
  const int b_lane = ((l31 >> 4) * kHalo + (l31 & 15)) * PSTR + 4 * hsel + (2 * wp * PB) * kHalo * PSTR;
  // A (weights): [g = hsel][o][4]
  const int a_lane = (hsel * NT + wn * (32 * MB) + l31) * 4;

  // ---- prologue: first tile's input ----
  {
    float v[ROUNDS];
    gather(tile_of(lid), v);
    scatter(in_s, v);
  }
  __syncthreads();

  const size_t img_pix = plane;
  // one 16-byte piece of a finished tile: register quad g of accumulator (mb, pb) = channels 8g + 4*hsel .. +3 of one
  // pixel (conv3x3_mfma_kernel's epilogue); j = 8*pb + 4*mb + g
  auto store_piece = [&](int j, const f32x16 (&h)[MB][PB], const Tile& t, bool valid) __attribute__((always_inline)) {
    const int pb = j >> 3, mb = (j >> 2) & 1, g = j & 3;
    const int blk = wp * PB + pb;
    const int y = t.ty0 + 2 * blk + (l31 >> 4);
    const int x = t.tx0 + (l31 & 15);
    if (valid && y < p.h && x < p.w) {
      const size_t pix = (size_t)t.img * img_pix + (size_t)y * p.w + x;
      const int c0 = slab * NT + wn * (32 * MB) + mb * 32 + 8 * g + 4 * hsel;
      f32x4 v = {h[mb][pb][4 * g], h[mb][pb][4 * g + 1], h[mb][pb][4 * g + 2], h[mb][pb][4 * g + 3]};
      v += *reinterpret_cast<const f32x4*>(bias_s + c0 - slab * NT);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      if constexpr ((ABL & 1) != 0) {
        asm volatile("" ::"v"(v));
      } else {
        static_assert(EPI == kEpiRelu, "fp32 NHWC output only: the plane-writing forms are conv3x3_first16.hip's");
        *reinterpret_cast<f32x4*>(p.out + pix * COUT + c0) = v;
      }
    }
  };

  // Rotated loop: iteration `it` first takes over tile it-1's accumulators (`held`), then computes tile `it` while
  // held's sixteen pieces leave two per tap; one extra iteration writes the last tile out at once.
  f32x16 acc[MB][PB], held[MB][PB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mb][pb][e] = 0.f;
  for (int it = 0; it <= my_items; ++it) {
    const Tile tprev = tile_of(it > 0 ? lid + (it - 1) * G : lid);
    const bool vprev = it > 0;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) held[mb][pb] = acc[mb][pb];
    if (it == my_items) {
#pragma unroll
      for (int j = 0; j < 16; ++j) store_piece(j, held, tprev, vprev);
      break;
    }
    const int item = lid + it * G;
    const float* const ib = in_s + (it & 1) * IN_FLOATS;
    float* const ib_next = in_s + ((it + 1) & 1) * IN_FLOATS;
    // the next tile's input: its loads fly under this tile's MFMAs (past the last item: this tile again, never read)
    float nv[ROUNDS];
    if constexpr (!(ABL & 4)) gather(tile_of(it + 1 < my_items ? item + G : item), nv);
    else for (int r = 0; r < ROUNDS; ++r) nv[r] = 1.f;

#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mb][pb][e] = 0.f;

#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - dy * 3;
      const float* const bp = ib + b_lane + (dy * kHalo + dx) * PSTR;
      const float* const ap = w_s + tap * WCH3 + a_lane;
      if constexpr (!(ABL & 2)) {
        // channels 0-7: MFMA j pairs channel j (lanes 0-31) with channel 4 + j (lanes 32-63)
        f32x4 a[MB], b[PB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) a[mb] = *reinterpret_cast<const f32x4*>(ap + (mb * 32) * 4);
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) b[pb] = *reinterpret_cast<const f32x4*>(bp + pb * 2 * kHalo * PSTR);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
              acc[mb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb][j], b[pb][j], acc[mb][pb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // channels 8 .. CREAL-1: lanes 0-31 supply channel 8 + 2m, lanes 32-63 channel 9 + 2m (both from k-group 2)
#pragma unroll
        for (int m = 0; m < (CREAL - 8) / 2; ++m) {
          float a1[MB], b1[PB];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) a1[mb] = ap[(2 * NT + mb * 32) * 4 - hsel * NT * 4 + 2 * m + hsel];
#pragma unroll
          for (int pb = 0; pb < PB; ++pb) b1[pb] = bp[pb * 2 * kHalo * PSTR + 8 - 4 * hsel + 2 * m + hsel];
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
              acc[mb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[mb], b1[pb], acc[mb][pb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // two pieces of the previous tile, issued behind this tap's MFMAs (a version that also read the operand fragments one
      // tap ahead and put the pieces between MFMA groups was slower: 125 vs 117 us)
      if (tap < 8) {
        store_piece(2 * tap, held, tprev, vprev);
        store_piece(2 * tap + 1, held, tprev, vprev);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // the next tile's halo into the other buffer (every wave read it for the last time one barrier ago)
    scatter(ib_next, nv);
    __syncthreads();
  }
}

template <int CREAL, int COUT, int EPI, int ABL = 0>
static hipError_t launch_first_one(const ConvParams& p, const FirstInputs& f, hipStream_t stream) {
  auto kern = conv3x3_first_kernel<CREAL, COUT, EPI, ABL>;
  static KernelOnce once;
  int cus = 0;
  hipError_t e = once.prepare(reinterpret_cast<const void*>(kern), first::LDS_BYTES, &cus);
  if (e != hipSuccess) return e;
  constexpr int NS = COUT / first::NT;
  const long long items = (long long)p.n * p.tiles_x * p.tiles_y * NS;
  if (items <= 0 || items > 0x7fffffffLL) return hipErrorInvalidValue;
  int grid = (int)(items < cus ? items : cus);
  grid -= grid % NS;                                   // a workgroup keeps one slab: item stride G must preserve item % NS
  if (grid < NS) grid = NS;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(first::THREADS), first::LDS_BYTES, stream, p, f, (int)items);
  return hipGetLastError();
}

// p.in = x10, p.aux = x20 (NCHW), p.wpk / p.bias: weights packed with PackGeom{16, 128, 16, cout, .}; p.out NHWC fp32
// (kEpiRelu only).  hipErrorNotSupported: channel counts
// other than 10 / 12 (the generic pack_inputs + conv3x3_mfma path handles those).
hipError_t launch_conv3x3_first(const ConvParams& p, const FirstInputs& f, int cout, int epilogue, hipStream_t stream, int ablate) {
  const int creal = f.c10 + f.c20 + f.c60;
  if (f.c10 != 4 || f.c20 != 6 || (f.c60 != 0 && f.c60 != 2)) return hipErrorNotSupported;      // the Sentinel-2 band groups
  if ((size_t)p.h * p.w * 6 * 4 >= ((size_t)1 << 31)) return hipErrorNotSupported;               // 32-bit offsets inside one image
  if (!p.in || !p.aux || (f.c60 > 0 && !f.x60) || !p.out) return hipErrorInvalidValue;
  if ((size_t)p.h * p.w * (size_t)cout * 4 >= ((size_t)1 << 40)) return hipErrorInvalidValue;
#define DSEN2_FIRST(CR, CO) \
  if (creal == CR && cout == CO) return launch_first_one<CR, CO, kEpiRelu>(p, f, stream);
  if (epilogue != kEpiRelu) return hipErrorInvalidValue;      // the (hi, lo) / (hi | xl, lo16) plane forms: conv3x3_first16.hip
#ifdef DSEN2_DIAG
  if (creal == 10 && cout == 128 && epilogue == kEpiRelu) {
    if (ablate == 1) return launch_first_one<10, 128, kEpiRelu, 1>(p, f, stream);
    if (ablate == 2) return launch_first_one<10, 128, kEpiRelu, 2>(p, f, stream);
    if (ablate == 4) return launch_first_one<10, 128, kEpiRelu, 4>(p, f, stream);
    if (ablate == 3) return launch_first_one<10, 128, kEpiRelu, 3>(p, f, stream);
    if (ablate == 5) return launch_first_one<10, 128, kEpiRelu, 5>(p, f, stream);
    if (ablate == 7) return launch_first_one<10, 128, kEpiRelu, 7>(p, f, stream);
    if (ablate == 6) return launch_first_one<10, 128, kEpiRelu, 6>(p, f, stream);
  }
#endif
  DSEN2_FIRST(10, 128) DSEN2_FIRST(12, 128) DSEN2_FIRST(10, 256) DSEN2_FIRST(12, 256)
#undef DSEN2_FIRST
  return hipErrorNotSupported;
}

}  // namespace dsen2
