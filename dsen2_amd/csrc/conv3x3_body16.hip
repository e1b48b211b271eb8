// conv3x3_body16.hip — bf16-operand F->F 3x3 'same' convolution built around v_mfma_f32_16x16x32_bf16.
//
// Same persistent, software-pipelined structure as conv3x3_body.hip (one 8-wave workgroup per CU walking
// (16x16 tile, 128-channel slab) items; 3-slot weight ring; double-buffered 64-channel input chunks; operand
// fragments double buffered in registers), re-cut for the 16x16x32 shape with the WEIGHTS as the A operand:
//
//   D[row = output channel][col = pixel] = sum_k W[row][k] * X[k][col]
//
//   * a lane then owns ONE pixel (col = lane & 15) and 4 output channels per accumulator (rows 4*(lane>>4) + r).
//     The weight packing permutes the rows (pack_conv_weights_bf16_host, perm16) so that the two accumulators
//     of a 32-channel pair hold 8 CONSECUTIVE channels of that pixel: the epilogue is one 16-byte bf16 store
//     (or two 16-byte fp32 stores + residual loads) per lane, pair and pixel row — no cross-lane transposes,
//     8 stores per lane and item where the 32x32 form needs 64;
//   * the pixel fragment is a 16-pixel row: with 40 words per halo pixel in LDS every ds_read_b128 of the
//     loop is bank-conflict free (the 2x16 block of the 32x32x16 form is 2-way on any stride);
//   * on random data the chip holds a higher clock on the 16x16x32 shape than on 32x32x16 at equal cycles per
//     FLOP (MI355X_MICROARCH.md, DVFS give-back item 7).
//
// Per wave: 64 channels x 64 pixels = 4 x 4 accumulators; one step = (tap, 64 input channels) = 2 k-steps of
// 16 MFMAs; 8 ds_read_b128 per k-step.
#include <type_traits>

#include "dsen2_internal.h"

namespace dsen2 {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 32;                      // 32-bit words per pixel and step = 64 bf16 channels
constexpr int NT = 128;                     // output channels per item
constexpr int THREADS = 512;
constexpr int PSTR = 40;                    // words per halo pixel in LDS: 16-pixel rows read conflict free
constexpr int IN_WORDS = kHaloPix * PSTR;
constexpr int WCH = KC * NT;                // words per weight chunk (16 KiB)
constexpr int QPP = KC / 4;                 // 16-byte pieces per pixel and chunk
constexpr int IN_PIECES = kHaloPix * QPP;
constexpr int IN_ROUNDS = (IN_PIECES + THREADS - 1) / THREADS;      // 6
constexpr int W_ROUNDS = (WCH / 4) / THREADS;                       // 2
constexpr int NWBUF = 3;
constexpr int KSTEPS = 2;                   // k = 32 channels per MFMA
constexpr int RS = 4;                       // tile rows per wave
constexpr int PB = 4, MB = 4;               // 16-pixel rows x 16-channel blocks per wave
constexpr size_t LDS_BYTES = (size_t)(2 * IN_WORDS + NWBUF * WCH + 256) * 4;      // + the bias vector
static_assert(IN_ROUNDS <= 8, "input round r is written mid tap r");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

}  // namespace

// CINW = words per input pixel (= F / 2).  ABL: timing-only ablation mask as in conv3x3_body.hip
// (1 no stores, 2 no residual loads, 4 no weight stream, 8 no input stream, 16 no barriers).
template <int CINW, int COUT, int EPI, int ABL>
__global__ __launch_bounds__(THREADS, 2) void conv3x3_body16_kernel(const ConvParams p, const int n_items) {
  constexpr int NCC = CINW / KC;
  constexpr int NCHUNK = NCC * 9;
  constexpr int NS = COUT / NT;
  static_assert(NCC % 2 == 0, "input double buffer parity");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const in_s = smem;                      // [2][324][PSTR]
  float* const w_s = smem + 2 * IN_WORDS;        // [3][8 k-groups][128 rows][4 words]
  float* const bias_s = w_s + NWBUF * WCH;       // [COUT]: the epilogue must not queue a vector-memory load
                                                 // behind its own stores (vmcnt retires in issue order)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1;                 // 64-channel half of the slab
  const int wp = wave >> 1;                // 4-row strip of the tile
  const int l15 = lane & 15;
  const int q4 = lane >> 4;

  // persistent schedule (XCD-contiguous logical ids), as conv3x3_body.hip
  const int G = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  if (lid >= n_items) return;
  const int my_items = (n_items - lid + G - 1) / G;
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const size_t img_pix = (size_t)p.h * p.w;

  // ---- input staging (branch-free loads; zero padding selected in at the LDS write) ----
  int g_off[IN_ROUNDS];
  int s_off[IN_ROUNDS];
  const float* stage_img = p.in;
#pragma unroll
  for (int r = 0; r < IN_ROUNDS; ++r) {
    const int piece = r * THREADS + tid;
    const int hp = piece / QPP, qq = piece - hp * QPP;
    s_off[r] = piece < IN_PIECES ? hp * PSTR + qq * 4 : -1;
  }
  auto set_stage_item = [&](int item) {
    const int tile = item / NS;
    const int img = tile / tiles_per_img;
    const int trem = tile - img * tiles_per_img;
    const int tyi = trem / p.tiles_x;
    const int ty0 = tyi * kTile, tx0 = (trem - tyi * p.tiles_x) * kTile;
    stage_img = p.in + (size_t)img * img_pix * CINW;
#pragma unroll
    for (int r = 0; r < IN_ROUNDS; ++r) {
      const int piece = r * THREADS + tid;
      const int hp = piece / QPP, qq = piece - hp * QPP;
      const int hy = hp / kHalo, hx = hp - hy * kHalo;
      const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
      const bool inb = piece < IN_PIECES && (unsigned)gy < (unsigned)p.h && (unsigned)gx < (unsigned)p.w;
      g_off[r] = inb ? (gy * p.w + gx) * CINW + qq * 4 : -1;
    }
  };
  auto load_in = [&](int r, int cc) -> f32x4 {
    return *reinterpret_cast<const f32x4*>(stage_img + (g_off[r] >= 0 ? g_off[r] : 0) + cc * KC);
  };
  auto store_in = [&](float* buf, int r, f32x4 t) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = g_off[r] >= 0 ? t[e] : 0.f;
    if (s_off[r] >= 0) *reinterpret_cast<f32x4*>(buf + s_off[r]) = v;
  };

  // ---- weight stream: chunk c loaded (global -> VGPR) in step c-3, written to ring slot c%3 in step c-2,
  //      first read by the fragment prefetch at the end of step c-1 ----
  int wl_item = lid;
  int wl_chunk = 0;
  int st_slot = 0;
  auto load_w = [&](f32x4 (&wr)[W_ROUNDS]) {
    const float* src = p.wpk + ((size_t)(wl_item % NS) * NCHUNK + wl_chunk) * WCH + tid * 4;
#pragma unroll
    for (int r = 0; r < W_ROUNDS; ++r) wr[r] = *reinterpret_cast<const f32x4*>(src + r * THREADS * 4);
    if (++wl_chunk == NCHUNK) {
      wl_chunk = 0;
      wl_item = wl_item + G < n_items ? wl_item + G : lid;
    }
  };
  auto store_w = [&](const f32x4 (&wr)[W_ROUNDS]) {
    float* dst = w_s + st_slot * WCH + tid * 4;
#pragma unroll
    for (int r = 0; r < W_ROUNDS; ++r) *reinterpret_cast<f32x4*>(dst + r * THREADS * 4) = wr[r];
    st_slot = st_slot == NWBUF - 1 ? 0 : st_slot + 1;
  };

  // ---- per-lane operand addresses (words) ----
  // B operand (pixels): lane -> pixel column l15 of a 16-pixel row, channels 8*q4 .. +7 of the k-step
  // A operand (weights): [k-group = 4*s + q4][row = wn*64 + 16*mb + l15][4 words]
  const int x_lane = ((RS * wp) * kHalo + l15) * PSTR + 4 * q4;
  const int w_lane = (q4 * NT + wn * 64 + l15) * 4;

  f32x4 wr[W_ROUNDS];
  f32x4 ir;
  set_stage_item(lid);
  {
    f32x4 ir0[IN_ROUNDS];
#pragma unroll
    for (int r = 0; r < IN_ROUNDS; ++r) ir0[r] = load_in(r, 0);
    f32x4 w0[W_ROUNDS], w1[W_ROUNDS];
    load_w(w0);
    load_w(w1);
    store_w(w0);
    store_w(w1);
#pragma unroll
    for (int r = 0; r < IN_ROUNDS; ++r) store_in(in_s, r, ir0[r]);
    load_w(wr);
    ir = load_in(0, 1);
    if (tid < COUT) bias_s[tid] = p.bias[tid];
  }
  __syncthreads();

  f32x4 w_cur[MB], x_cur[PB], w_nxt[MB], x_nxt[PB];
  int mf_slot = 0;
  auto read_frags = [&](f32x4 (&wf)[MB], f32x4 (&xf)[PB], const float* ib, const float* wb, int tap, int s) {
    const int dy = tap / 3, dx = tap - dy * 3;
    const float* wp_ = wb + w_lane + (4 * s * NT) * 4;
    const float* xp_ = ib + x_lane + (dy * kHalo + dx) * PSTR + 16 * s;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) wf[mb] = *reinterpret_cast<const f32x4*>(wp_ + mb * 16 * 4);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) xf[pb] = *reinterpret_cast<const f32x4*>(xp_ + pb * kHalo * PSTR);
  };
  read_frags(w_cur, x_cur, in_s, w_s, 0, 0);

  // The item loop is ROTATED: an iteration first writes out the PREVIOUS item's accumulators, then runs this
  // item's 36 steps (one extra iteration writes the last item; the first one stores to out-of-range offsets,
  // which the buffer unit drops).  The staging loads issued in an item's last step are consumed half a step into
  // the next item, i.e. right after the epilogue's stores, and vmcnt retires in issue order: with every path
  // into the first step's wait passing the same stores, hipcc emits vmcnt(#stores + k) there and the stores
  // retire behind the next item's MFMAs; with the epilogue at the loop's tail the wait merges with the
  // prologue's store-free path, becomes vmcnt(k) and drains the stores at the head of every item.  For the
  // same reason the first input chunk (cc = 0) is its own copy of the step code.
  auto run = [&](auto mid_c) {
  constexpr int MIDS = decltype(mid_c)::value;
  f32x4 acc[MB][PB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) acc[mb][pb] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int it = 0; it <= my_items; ++it) {
    // ---- epilogue of item it-1: lane = pixel (ex, ey + pb), 8 consecutive channels ch8 + 32*pr per pair ----
    {
      const bool valid = it > 0;
      const int item = valid ? lid + (it - 1) * G : lid;
      const int tile = item / NS, slab = item - tile * NS;
      const int img = tile / tiles_per_img;
      const int trem = tile - img * tiles_per_img;
      const int tyi = trem / p.tiles_x;
      const int ty0 = tyi * kTile, tx0 = (trem - tyi * p.tiles_x) * kTile;
      // one buffer descriptor per image; pixels outside a ragged tile get an out-of-range offset, not a branch
      constexpr unsigned OB = EPI == kEpiRelu ? 2u : 4u;          // bytes per element of p.out
      const int ch8 = slab * NT + wn * 64 + 8 * q4;
      const auto aux_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(p.aux) + (EPI == kEpiResidual ? (size_t)img * img_pix * COUT : 0), 0,
          EPI == kEpiResidual ? (unsigned)(img_pix * COUT * 4) : 0, 0x00020000);
      const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<char*>(p.out) + (size_t)img * img_pix * COUT * OB, 0, (unsigned)(img_pix * COUT * OB), 0x00020000);
      const auto out2_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<char*>(p.out2) + (EPI == kEpiResidual ? (size_t)img * img_pix * COUT * 2 : 0), 0,
          EPI == kEpiResidual ? (unsigned)(img_pix * COUT * 2) : 0, 0x00020000);
      const int ex = tx0 + l15, ey = ty0 + RS * wp;
      const unsigned lane_eoff = (unsigned)((ey * p.w + ex) * COUT + ch8);              // elements
      const bool col_ok = valid && ex < p.w;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias_s + ch8 + 32 * pr);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias_s + ch8 + 32 * pr + 4);
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
          f32x4 v0 = acc[2 * pr][pb] + b0, v1 = acc[2 * pr + 1][pb] + b1;
          // elements; 0x20000000 is out of range once scaled to bytes (x2, x4)
          const unsigned eo = col_ok && ey + pb < p.h ? lane_eoff + (unsigned)(pb * p.w * COUT + pr * 32) : 0x20000000u;
          if constexpr (EPI == kEpiRelu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v0[e] = fmaxf(v0[e], 0.f);
              v1[e] = fmaxf(v1[e], 0.f);
            }
            const u32x4 h = {pack_bf16(v0[0], v0[1]), pack_bf16(v0[2], v0[3]), pack_bf16(v1[0], v1[1]), pack_bf16(v1[2], v1[3])};
            if constexpr (!(ABL & 1))
              __builtin_amdgcn_raw_buffer_store_b128(h, out_rsrc, eo * 2u, 0, 0);
            else
              asm volatile("" ::"v"(h));
          } else {
            f32x4 r0 = {1.f, 1.f, 1.f, 1.f}, r1 = r0;
            if constexpr (!(ABL & 2)) {
              r0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(aux_rsrc, eo * 4u, 0, 0));
              r1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(aux_rsrc, eo * 4u + 16u, 0, 0));
            }
            v0 = r0 + v0 * p.res_scale;       // -ffp-contract=off: two roundings, as keras
            v1 = r1 + v1 * p.res_scale;
            const u32x4 h = {pack_bf16(v0[0], v0[1]), pack_bf16(v0[2], v0[3]), pack_bf16(v1[0], v1[1]), pack_bf16(v1[2], v1[3])};
            if constexpr (!(ABL & 1)) {
              // immediate soffset only: see the store-data hazard note in conv3x3_body.hip
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v0), out_rsrc, eo * 4u, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v1), out_rsrc, eo * 4u + 16u, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b128(h, out2_rsrc, eo * 2u, 0, 0);
            } else {
              asm volatile("" ::"v"(v0), "v"(v1), "v"(h));
            }
          }
        }
      }
    }
    if (it == my_items) break;
    const int item = lid + it * G;
    const bool have_next_item = it + 1 < my_items;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) acc[mb][pb] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto do_cc = [&](const int cc, auto first_c) {
      (void)first_c;     // distinct instantiation = distinct copy of the step code for cc == 0
      const float* const ib = in_s + (cc & 1) * IN_WORDS;
      float* const ib_next = in_s + ((cc + 1) & 1) * IN_WORDS;
      const bool last_cc = cc == NCC - 1;
      const int in_cc = last_cc ? 0 : cc + 1;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float* const wb = w_s + mf_slot * WCH;
        const int nx_slot = mf_slot == NWBUF - 1 ? 0 : mf_slot + 1;
        const float* const wb_nx = w_s + nx_slot * WCH;

#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
          if (s < KSTEPS - 1) {
            read_frags(w_nxt, x_nxt, ib, wb, tap, s + 1);
          } else if (tap < 8) {
            read_frags(w_nxt, x_nxt, ib, wb_nx, tap + 1, 0);
          } else {
            read_frags(w_nxt, x_nxt, ib_next, wb_nx, 0, 0);
          }
          if (s == MIDS) {
            if constexpr (!(ABL & 4)) store_w(wr);
            if constexpr (!(ABL & 8))
              if (tap < IN_ROUNDS) store_in(ib_next, tap < IN_ROUNDS ? tap : 0, ir);
            if constexpr (!(ABL & 4)) load_w(wr);
            if constexpr (!(ABL & 8)) {
              if (tap + 1 < IN_ROUNDS) {
                ir = load_in(tap + 1 < IN_ROUNDS ? tap + 1 : 0, in_cc);
              } else if (tap == 8) {
                const int nn = cc + 2;
                if (nn == NCC && have_next_item) set_stage_item(item + G);
                ir = load_in(0, nn < NCC ? nn : nn - NCC);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
              acc[mb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w_cur[mb]),
                                                                    __builtin_bit_cast(bf16x8, x_cur[pb]),
                                                                    acc[mb][pb], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < MB; ++q) w_cur[q] = w_nxt[q];
#pragma unroll
          for (int q = 0; q < PB; ++q) x_cur[q] = x_nxt[q];
        }
        mf_slot = nx_slot;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 16)) __syncthreads();
      }
    };
    do_cc(0, std::true_type{});
#pragma unroll 1
    for (int cc = 1; cc < NCC; ++cc) do_cc(cc, std::false_type{});
  }
  };   // run
  // (a wave-group stagger of the staging slot, as in conv3x3_body.hip, needs two copies of the loop: 33 spills)
  run(std::integral_constant<int, 0>{});
}

template <int CINW, int COUT, int EPI, int ABL = 0>
static hipError_t launch_body16_one(const ConvParams& p, hipStream_t stream) {
  auto kern = conv3x3_body16_kernel<CINW, COUT, EPI, ABL>;
  static bool attr_set[64] = {};
  static int cus[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  // per-image buffer descriptors address elements with 32 bits
  if ((size_t)p.h * p.w * COUT * 4 >= 0x40000000ull) return hipErrorInvalidValue;
  const long long items = (long long)p.n * p.tiles_x * p.tiles_y * (COUT / NT);
  if (items <= 0 || items > 0x7fffffffLL) return hipErrorInvalidValue;
  const int grid = (int)(items < cus[dev] ? items : cus[dev]);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), LDS_BYTES, stream, p, (int)items);
  return hipGetLastError();
}

hipError_t launch_conv3x3_body16(const ConvParams& p, int feat, int epilogue, hipStream_t stream) {
  if (epilogue == kEpiResidual && !p.out2) return hipErrorInvalidValue;
  if (feat != 256) return hipErrorInvalidValue;     // F = 128 (two input chunks) compiles to 58 spills: not offered
  if (g_body_ablate != 0) {
#define DSEN2_ABL(M)                                                                     \
  if (g_body_ablate == M)                                                                \
    return epilogue == kEpiRelu ? launch_body16_one<128, 256, kEpiRelu, M>(p, stream)    \
                                : launch_body16_one<128, 256, kEpiResidual, M>(p, stream);
    DSEN2_ABL(1) DSEN2_ABL(3) DSEN2_ABL(4) DSEN2_ABL(8) DSEN2_ABL(12) DSEN2_ABL(15) DSEN2_ABL(16) DSEN2_ABL(31)
#undef DSEN2_ABL
    return hipErrorInvalidValue;
  }
  return epilogue == kEpiRelu ? launch_body16_one<128, 256, kEpiRelu>(p, stream)
                              : launch_body16_one<128, 256, kEpiResidual>(p, stream);
}

}  // namespace dsen2
