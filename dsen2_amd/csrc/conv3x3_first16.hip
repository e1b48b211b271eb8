// conv3x3_first16.hip — the network's FIRST convolution (utils/DSen2Net.py:24-29: Concatenate(axis=1) + Conv2D(F, 3x3,
// 'same') + bias + ReLU) for the bf16-operand modes (precision 1: bf16 operands; precision 2: bf16x3), on the bf16
// matrix cores.
//
// Why a second kernel.  conv3x3_first.hip contracts K = 9 taps x 10 (12) channels with v_mfma_f32_32x32x2_f32: 12 GFLOP =
// 77 us of fp32 matrix-core time at the bench batch, 117-146 us measured — next to residual blocks that take 1.5 ms
// (precision 1) or 3.9 ms (precision 2) for the whole network that was 13 % / 6 % of a step, spent on an arithmetic
// the rest of the step no longer uses.  With v_mfma_f32_32x32x16_bf16 a tap is ONE instruction per 32 x 32 output block
// (the 10 / 12 channels in a 16-slot K, the unused slots zero on both sides): 9 (x 3) MFMAs per block and tile instead of
// 45, each at 16 x the fp32 rate — the layer becomes what it should be, a streaming WRITE of the residual stream's planes
// (268 MB at the bench batch; precision 2: 402 MB), with the gather of the 21 MB of inputs and the MFMAs underneath:
// 139.5 -> 59 us (precision 2: 161 -> 90 us), store-bound at 4.6 TB/s (profiles/r05_first16_ab.txt).
//
// Structure = conv3x3_first.hip's, operand format aside:
//   * persistent, one workgroup of 8 waves per CU walks tiles lid, lid + G, ... and keeps ONE 128-channel output slab:
//     its weights ([plane][tap][k half][o: 128][8 bf16] — the packed buffer is the LDS image) and bias go to LDS once;
//   * the halo tile of the NEXT item is gathered from the NCHW inputs into registers (per-image buffer descriptors; the
//     Concatenate is an address computation) while the current tile computes, then converted to bf16 and written into
//     the other half of a double buffer as [halo pixel][16 channel slots] with a 48-byte pixel pitch (conflict-free
//     ds_read_b128 over 16 pixels); precision 2 writes two planes, xh = bf16(x) and xl = bf16(x - xh);
//   * the epilogue is DEFERRED: a finished tile's accumulators are written out inside the next tile's tap loop, two 16-byte
//     pieces per tap — as ONE 16-byte buffer store per lane and plane (v_permlane32_swap pairs the half-waves' quads), issued by
//     every lane through per-image descriptors (out-of-range offsets instead of branches: the loop body is straight-line code).
// Arithmetic.  precision 1: out = relu(sum_k bf16(x_k) * bf16(w_k) + b), every product exact in fp32, fp32 accumulate.
// precision 2: x = xh + xl, w = wh + wl (weights split at pack time), a product = xh*wh + xh*wl + xl*wh in fp32 — the
// same three-MFMA form as the residual blocks (conv3x3_body16w.hip, X3); the xl*wl term is 2^-18 of the product.
// The output is the residual stream in the form the body kernels read: precision 1 the blocked (hi, lo) planes, precision
// 2 hx = (hi | xl) planes + lo16 (what launch_split_f32 / launch_split3_f32 make of an fp32 tensor).
// Parity: tests/test_gpu_first16.py (against a float64 restatement on the operands the kernel multiplies; ragged shapes;
// DSen2_60's 12 channels) and the whole-network gates of tests/test_gpu_bf16.py / test_gpu_bf16x3.py.
#include <string.h>

#include "conv3x3_bf16_common.h"
#include "dsen2_internal.h"

// Compile-time switches of VARIANT builds only (python -m dsen2_amd.build --variant NAME -D...; tools/ab_first16.sh): the
// product is built with the default.  FIRST16_ABL = timing-only ablation mask (1 no stores, 2 no MFMAs, 4 no input gather;
// outputs are wrong).
#ifndef FIRST16_ABL
#define FIRST16_ABL 0
#endif

namespace dsen2 {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
using bf16k::bf16x8;
using bf16k::u32x4;

namespace first16 {
constexpr int THREADS = 512;                        // 8 waves: 2 (64-channel halves) x 4 (pixel quarters: 4 rows x 16)
constexpr int NT = 128;                             // output channels per item (slab)
constexpr int KSLOTS = 16;                          // channel slots of one MFMA (K of v_mfma_f32_32x32x16_bf16)
constexpr int PITCH = 24;                           // bf16 per halo pixel in LDS (48 B: 16 slots + padding against bank conflicts)
constexpr int IN_PLANE_BYTES = kHaloPix * PITCH * 2;              // 15,552
constexpr int W_PLANE_U16 = 9 * 2 * NT * 8;                       // one slab, one weight plane: [tap][k half][o][8] = 18,432 bf16
constexpr int W_PLANE_BYTES = W_PLANE_U16 * 2;                    // 36,864
constexpr size_t lds_bytes(bool x3) { return (size_t)(x3 ? 2 : 1) * (W_PLANE_BYTES + 2 * IN_PLANE_BYTES) + NT * sizeof(float); }
static_assert(lds_bytes(true) <= 160 * 1024, "LDS budget");
}  // namespace first16

}  // namespace

// CREAL: real input channels (10 = 4 + 6 or 12 = 4 + 6 + 2).  X3 = false: precision 1, p.out / p.out2 = the blocked (hi, lo)
// planes; X3 = true: precision 2, p.out = hx (two planes per image: hi | xl), p.out2 = lo16.  p.in = x10, p.aux = x20 (NCHW);
// x60 and the channel counts come in `f`; p.wpk = pack_first16_weights_host's buffer.
template <int CREAL, int COUT, bool X3>
__global__ __launch_bounds__(first16::THREADS, 2) void conv3x3_first16_kernel(const ConvParams p, const FirstInputs f, const int n_items) {
  using namespace first16;
  constexpr int NS = COUT / NT;
  constexpr int MB = 2, PB = 2;
  constexpr int PL = X3 ? 2 : 1;                    // operand planes on each side
  static_assert(CREAL % 2 == 0 && CREAL > 8 && CREAL <= KSLOTS, "first-layer form");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const w_s = smem;                                               // [PL][9][2][128][8] bf16
  char* const in_s = smem + PL * W_PLANE_BYTES;                         // [2 buffers][PL][324][PITCH] bf16
  float* const bias_s = reinterpret_cast<float*>(in_s + 2 * PL * IN_PLANE_BYTES);      // [128]

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

  // ---- weights of this slab (all planes): global -> LDS once; both input buffers zeroed (the unused channel slots of a
  // halo pixel are read by the MFMAs: they must be zeros, not whatever the LDS held) ----
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.wpk) + (size_t)slab * PL * W_PLANE_BYTES);
    u32x4* dst = reinterpret_cast<u32x4*>(w_s);
    for (int i = tid; i < PL * W_PLANE_BYTES / 16; i += THREADS) dst[i] = src[i];
    u32x4* z = reinterpret_cast<u32x4*>(in_s);
    for (int i = tid; i < 2 * PL * IN_PLANE_BYTES / 16; i += THREADS) z[i] = u32x4{0u, 0u, 0u, 0u};
    if (tid < NT) bias_s[tid] = p.bias[slab * NT + tid];
  }
  __syncthreads();

  // ---- gather geometry of one halo tile (conv3x3_first.hip's): per input tensor its CT x 324 values in rounds of 512
  // threads, channel outer / halo pixel inner.  pk = hx | hy << 5 | (channel inside its tensor) << 10 | (LDS bf16 index of
  // the value inside a plane) << 13 | (no element) << 30.  A lane without an element still loads (out of range: zero) and
  // still writes — into the padding slots 16-23 of a halo pixel, which no operand read covers — so that the loop body has NO
  // divergent branch: hipcc then counts the vector-memory operations in flight exactly (see the waits below) ----
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
        pk[r0 + r] = have ? hx | hy << 5 | c << 10 | (hp * PITCH + cbase + c) << 13
                          : (hp * PITCH + KSLOTS + (c & 7)) << 13 | 1 << 30;
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
        const bool inb = (k >> 30) == 0 && (unsigned)gy < (unsigned)p.h && (unsigned)gx < (unsigned)p.w;
        const unsigned voff = inb ? (unsigned)((((k >> 10) & 7) * (int)plane + gy * p.w + gx) * 4) : 0x80000000u;
        v[r0 + r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, 0, 0));
      }
    };
    fetch(0, R10, p.in, C10);
    fetch(R10, R20, p.aux, C20);
    if constexpr (R60 > 0) fetch(R10 + R20, R60, f.x60, C60);
  };
  // fp32 -> bf16 (RNE; precision 2: also the remainder's bf16) into one input buffer (PL planes, IN_PLANE_BYTES apart)
  auto scatter = [&](char* buf, const float (&v)[ROUNDS]) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const __bf16 xh = (__bf16)v[r];
      __bf16* const dst = reinterpret_cast<__bf16*>(buf) + ((pk[r] >> 13) & 0x1fff);
      dst[0] = xh;
      if constexpr (X3) dst[IN_PLANE_BYTES / 2] = (__bf16)(v[r] - (float)xh);
    }
  };

  // ---- per-lane operand byte offsets ----
  // B (pixels): lane -> pixel (row l31 >> 4, column l31 & 15) of a 2 x 16 block, channel slots 8 * hsel .. + 7
  const int b_lane = (((l31 >> 4) + 2 * wp * PB) * kHalo + (l31 & 15)) * (PITCH * 2) + 16 * hsel;
  // A (weights): [tap][k half = hsel][o][8]
  const int a_lane = (hsel * NT + wn * (32 * MB) + l31) * 16;

  // ---- prologue: first tile's input ----
  {
    float v[ROUNDS];
    gather(tile_of(lid), v);
    scatter(in_s, v);
  }
  __syncthreads();

  const size_t img_pix = plane;
  // Two 16-byte pieces of a finished tile (register quads g0 = even and g0 + 1 of one accumulator (mb, pb); tp = 4*pb + 2*mb
  // + g0/2) as ONE 16-byte store per lane and plane.  A lane holds 4 of a pixel's 8 channels of a block (quad g = channels 8g +
  // 4*hsel .. + 3), lane ^ 32 the other 4: v_permlane32_swap exchanges quad g0 of the upper half-wave with quad g0 + 1 of the
  // lower one, after which lanes 0-31 own the whole piece of block g0 and lanes 32-63 that of block g0 + 1 (blocked planes
  // [n][C/8][h][w][8], conv3x3_body16w.hip).  The stores go through per-IMAGE buffer descriptors (`so`: one per plane tensor)
  // and are issued by EVERY lane, always: a lane without a pixel (ragged tile edge, or the dummy pass before the first tile)
  // stores at an out-of-range offset, which the hardware drops — no divergent branch in the loop body.
  struct StoreRsrc { __amdgpu_buffer_rsrc_t out, out2; };
  auto store_rsrc = [&](const Tile& t) __attribute__((always_inline)) -> StoreRsrc {
    const size_t plane_bytes = (size_t)(COUT / 8) * img_pix * 16;      // one image, one plane: < 2^31 (launcher)
    const int img = __builtin_amdgcn_readfirstlane(t.img);
    return StoreRsrc{__builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.out) + (size_t)img * PL * plane_bytes, 0,
                                                       (unsigned)(PL * plane_bytes), 0x00020000),
                     __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.out2) + (size_t)img * plane_bytes, 0,
                                                       (unsigned)plane_bytes, 0x00020000)};
  };
  auto store_pair = [&](int tp, const f32x16 (&h)[MB][PB], const Tile& t, const StoreRsrc& so, bool valid) __attribute__((always_inline)) {
    const int j0 = 2 * tp;
    const int pb = j0 >> 3, mb = (j0 >> 2) & 1, g0 = j0 & 3;
    const int blk = wp * PB + pb;
    const int y = t.ty0 + 2 * blk + (l31 >> 4);
    const int x = t.tx0 + (l31 & 15);
    unsigned hh[2][2], ll[2][2], xx[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int g = g0 + q;
      const int cl = wn * (32 * MB) + mb * 32 + 8 * g + 4 * hsel;       // channel inside the slab
      f32x4 v = {h[mb][pb][4 * g], h[mb][pb][4 * g + 1], h[mb][pb][4 * g + 2], h[mb][pb][4 * g + 3]};
      v += *reinterpret_cast<const f32x4*>(bias_s + cl);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      bf16k::split2(__float_as_uint(v[0]), __float_as_uint(v[1]), hh[q][0], ll[q][0]);
      bf16k::split2(__float_as_uint(v[2]), __float_as_uint(v[3]), hh[q][1], ll[q][1]);
      if constexpr (X3) {
        xx[q][0] = bf16k::pack_bf16(v[0] - __uint_as_float(hh[q][0] << 16), v[1] - __uint_as_float(hh[q][0] & 0xffff0000u));
        xx[q][1] = bf16k::pack_bf16(v[2] - __uint_as_float(hh[q][1] << 16), v[3] - __uint_as_float(hh[q][1] & 0xffff0000u));
      }
    }
    // (a, b) -> lanes 0-31: (a, a of lane + 32); lanes 32-63: (b of lane - 32, b): with a = quad g0's register and b = quad
    // g0 + 1's, every lane ends up with channels 0-3 | 4-7 of ITS block (g0 + hsel)
    auto piece = [&](const unsigned (&r)[2][2]) __attribute__((always_inline)) -> u32x4 {
      const auto s01 = __builtin_amdgcn_permlane32_swap(r[0][0], r[1][0], false, false);
      const auto s23 = __builtin_amdgcn_permlane32_swap(r[0][1], r[1][1], false, false);
      return u32x4{s01[0], s23[0], s01[1], s23[1]};
    };
    const u32x4 ph = piece(hh), pl = piece(ll);
    u32x4 px = u32x4{0u, 0u, 0u, 0u};
    if constexpr (X3) px = piece(xx);
    if constexpr ((FIRST16_ABL & 1) != 0) {
      asm volatile("" ::"v"(ph), "v"(pl), "v"(px));
    } else {
      const int cb = (slab * NT + wn * (32 * MB) + mb * 32) / 8 + g0 + hsel;      // this lane's 8-channel block
      const bool have = valid && y < p.h && x < p.w;
      // 32-bit arithmetic: every offset inside one image's plane is below 2^31 (launcher); a select, not a branch
      const unsigned off = have ? ((unsigned)cb * (unsigned)img_pix + (unsigned)(y * p.w + x)) * 16u : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b128(ph, so.out, (int)off, 0, 0);
      // hx: plane 0 = hi, plane 1 = xl = bf16(x - hi), one plane further in the same image.  The plane offset goes into the
      // VECTOR offset (one add), soffset stays the immediate 0: a 128-bit buffer store with a REGISTER soffset reads its data
      // late on gfx950 and hipcc does not pad that hazard (experiments/README.md, round 1).  An out-of-range lane stays out
      // of range: 2^31 + a plane (< 2^30) does not wrap.
      if constexpr (X3)
        __builtin_amdgcn_raw_buffer_store_b128(px, so.out, (int)(off + (unsigned)((size_t)(COUT / 8) * img_pix * 16)), 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(pl, so.out2, (int)off, 0, 0);
    }
  };

  // Rotated loop: iteration `it` first takes over tile it-1's accumulators (`held`), then computes tile `it` while
  // held's sixteen pieces leave two per tap; one extra iteration writes the last tile out at once.
  // What bounds it (timing-only variant builds, tools/ab_first16.sh, profiles/r05_first16_ab.txt; DSen2_20, batch 512): the
  // STORES.  Without stores the kernel takes 36 us (bf16x3: 64), without MFMAs 54 us (71), complete 58-59 us (90): 268 MB
  // (402 MB) of plane writes at 4.6 TB/s (4.5), against 47 us measured for the same bytes as fp32 stores alone in round 3.  A
  // gather two tiles ahead (two register sets alternating over a twice-unrolled loop: exact counted waits, no register moves)
  // was built and measured: 58.0 us against 58.7 — the gather's latency is not what is exposed; left at one tile ahead.
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
    const StoreRsrc sprev = store_rsrc(tprev);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) held[mb][pb] = acc[mb][pb];
    if (it == my_items) {
#pragma unroll
      for (int tp = 0; tp < 8; ++tp) store_pair(tp, held, tprev, sprev, vprev);
      break;
    }
    const int item = lid + it * G;
    const char* const ib = in_s + (it & 1) * (PL * IN_PLANE_BYTES);
    char* const ib_next = in_s + ((it + 1) & 1) * (PL * IN_PLANE_BYTES);
    // the next tile's input: its loads fly under this tile's MFMAs and stores (past the last item: this tile again, never read)
    float nv[ROUNDS];
    if constexpr ((FIRST16_ABL & 4) == 0) gather(tile_of(it + 1 < my_items ? item + G : item), nv);
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
      const char* const bp = ib + b_lane + (dy * kHalo + dx) * (PITCH * 2);
      const char* const ap = w_s + tap * (2 * NT * 16) + a_lane;
      bf16x8 a[PL][MB], b[PL][PB];
#pragma unroll
      for (int q = 0; q < PL; ++q) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) a[q][mb] = *reinterpret_cast<const bf16x8*>(ap + q * W_PLANE_BYTES + (mb * 32) * 16);
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) b[q][pb] = *reinterpret_cast<const bf16x8*>(bp + q * IN_PLANE_BYTES + pb * 2 * kHalo * (PITCH * 2));
      }
      // precision 1: x * w.  precision 2: xh*wh + xh*wl + xl*wh (operand planes 0 = hi, 1 = lo)
#pragma unroll
      for (int term = 0; term < ((FIRST16_ABL & 2) ? 0 : X3 ? 3 : 1); ++term) {
        const int qa = term == 1 ? 1 : 0, qb = term == 2 ? 1 : 0;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int pb = 0; pb < PB; ++pb)
            acc[mb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[X3 ? qa : 0][mb], b[X3 ? qb : 0][pb], acc[mb][pb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // two pieces of the previous tile, issued behind this tap's MFMAs
      if (tap < 8) {
        store_pair(tap, held, tprev, sprev, vprev);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // the next tile's halo into the other buffer (every wave read it for the last time one barrier ago); the loop body has no
    // divergent branch, so hipcc's waits here are exact: vmcnt(16 stores + the younger loads), not a drain of the stores
    scatter(ib_next, nv);
    __syncthreads();
  }
}

template <int CREAL, int COUT, bool X3>
static hipError_t launch_first16_one(const ConvParams& p, const FirstInputs& f, hipStream_t stream) {
  auto kern = conv3x3_first16_kernel<CREAL, COUT, X3>;
  static KernelOnce once;
  int cus = 0;
  hipError_t e = once.prepare(reinterpret_cast<const void*>(kern), first16::lds_bytes(X3), &cus);
  if (e != hipSuccess) return e;
  constexpr int NS = COUT / first16::NT;
  const long long items = (long long)p.n * p.tiles_x * p.tiles_y * NS;
  if (items <= 0 || items > 0x7fffffffLL) return hipErrorInvalidValue;
  int grid = (int)(items < cus ? items : cus);
  grid -= grid % NS;                                   // a workgroup keeps one slab: item stride G must preserve item % NS
  if (grid < NS) grid = NS;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(first16::THREADS), first16::lds_bytes(X3), stream, p, f, (int)items);
  return hipGetLastError();
}

size_t first16_weight_u16(int cout, bool x3) { return (size_t)(cout / first16::NT) * (x3 ? 2 : 1) * first16::W_PLANE_U16; }

static inline uint16_t first16_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// kernel HWIO fp32 (3, 3, cin <= 16, cout) -> [slab][plane: wh (| wl)][tap][k half][o: 128][8] bf16, zero in the unused
// channel slots; wh = RNE bf16 of w, wl = RNE bf16 of w - wh (x3 only) — pack_conv_weights_bf16x3_host's split
void pack_first16_weights_host(const float* k, int cin, int cout, bool x3, uint16_t* dst) {
  const int nslab = cout / first16::NT, planes = x3 ? 2 : 1;
  size_t i = 0;
  for (int slab = 0; slab < nslab; ++slab)
    for (int q = 0; q < planes; ++q)
      for (int tap = 0; tap < 9; ++tap)
        for (int kh = 0; kh < 2; ++kh)
          for (int o = 0; o < first16::NT; ++o)
            for (int j = 0; j < 8; ++j, ++i) {
              const int c = 8 * kh + j, oc = slab * first16::NT + o;
              uint16_t v = 0;
              if (c < cin) {
                const float w = k[((size_t)tap * cin + c) * cout + oc];
                const uint16_t hb = first16_bf16_rne(w);
                if (q == 0) {
                  v = hb;
                } else {
                  const uint32_t hu = (uint32_t)hb << 16;
                  float wh;
                  memcpy(&wh, &hu, 4);
                  v = first16_bf16_rne(w - wh);
                }
              }
              dst[i] = v;
            }
}

// p.in = x10, p.aux = x20 (NCHW); p.wpk = pack_first16_weights_host(.., x3); p.bias fp32 [cout].  x3 = false: p.out / p.out2 =
// the blocked (hi, lo) planes; x3 = true: p.out = hx (hi | xl planes), p.out2 = lo16.  hipErrorNotSupported: channel counts
// other than 4 + 6 (+ 2) (the generic pack_inputs + conv3x3_mfma path handles those).
hipError_t launch_conv3x3_first16(const ConvParams& p, const FirstInputs& f, int cout, bool x3, hipStream_t stream) {
  const int creal = f.c10 + f.c20 + f.c60;
  if (f.c10 != 4 || f.c20 != 6 || (f.c60 != 0 && f.c60 != 2)) return hipErrorNotSupported;      // the Sentinel-2 band groups
  if ((size_t)p.h * p.w * 6 * 4 >= ((size_t)1 << 31)) return hipErrorNotSupported;               // 32-bit offsets inside one image
  if (!p.in || !p.aux || (f.c60 > 0 && !f.x60) || !p.out || !p.out2 || !p.wpk || !p.bias) return hipErrorInvalidValue;
  if ((size_t)p.h * p.w * (size_t)cout * 4 >= ((size_t)1 << 40)) return hipErrorInvalidValue;
#define DSEN2_FIRST16(CR, CO)                                                                               \
  if (creal == CR && cout == CO) return x3 ? launch_first16_one<CR, CO, true>(p, f, stream) : launch_first16_one<CR, CO, false>(p, f, stream);
  DSEN2_FIRST16(10, 128) DSEN2_FIRST16(12, 128) DSEN2_FIRST16(10, 256) DSEN2_FIRST16(12, 256)
#undef DSEN2_FIRST16
  return hipErrorNotSupported;
}

}  // namespace dsen2
