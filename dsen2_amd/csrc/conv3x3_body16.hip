// conv3x3_body16.hip — bf16-operand 256->256 3x3 'same' convolution: v_mfma_f32_16x16x32_bf16 fed by LDS-DMA.
//
// Same persistent structure as conv3x3_body.hip (one 8-wave workgroup per CU walking (16x16 tile, 128-channel
// slab) items, one step = (tap, 64 input channels), operand fragments double buffered in registers), with
// three changes that the bf16 rate makes necessary (profiles/r01_ablation.md):
//
//   1. STAGING BY LDS-DMA (conv3x3_dma.h, which also states the synchronisation rules).  Weight chunks (16 KiB per
//      step) and input chunks go global -> LDS with `buffer_load_dwordx4 ... lds`: no staging VGPRs, no ds_write, no
//      select; out-of-range lanes write the zero padding.
//   2. WEIGHTS ARE THE A OPERAND:  D[row = output channel][col = pixel].  A lane owns ONE pixel (col = lane & 15)
//      and 4 output channels per accumulator (rows 4*(lane>>4) + r); the weight packing permutes the rows
//      (pack_conv_weights_bf16_host, perm16) so that the two accumulators of a 32-channel pair hold 8
//      CONSECUTIVE channels: the epilogue is one 16-byte bf16 store (or two 16-byte fp32 stores + residual
//      loads) per lane, pair and pixel row — no cross-lane transposes, 8 stores per lane and item, not 64.
//   3. CONFLICT-FREE FRAGMENT READS.  The input chunk lives in LDS as [channel group q: 8][pixel slot: 336][16 B]
//      (the DMA's lane-linear destination makes any layout free): a pixel fragment is 16 consecutive pixels of
//      one group = 16 consecutive 16-byte slots, and 336 = 0 mod 16 keeps the two groups of a ds_read_b128 lane
//      group on disjoint banks.  The 2x16-pixel block of the 32x32x16 form is 2-way conflicted on any stride.
//
// Per wave: 64 channels x 64 pixels = 4 x 4 accumulators, 2 k-steps of 16 MFMAs and 8 ds_read_b128 per step.
#include <type_traits>

#include "conv3x3_dma.h"

namespace dsen2 {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Cache policy of conv-B's fp32 traffic: the residual stream is read once and written once per block and is as
// large as the Infinity Cache; `nt` (aux bit 1) on those loads and stores keeps them from evicting the weights and
// the bf16 activations that ARE re-read (VDSen2 bf16 bench, same box: 13.70 k -> 13.95 k patches/s; on the stores
// alone +0.5 %).  The bf16 copy, the next convolution's input, stays on the default policy.
constexpr int kAuxNt = 2;

using dma::KC; using dma::NT; using dma::THREADS; using dma::QS; using dma::IN_BYTES; using dma::IN_BLOCKS;
using dma::WCH; using dma::NWBUF; using dma::LDS_BYTES; using dma::wait_vmcnt;
constexpr int KSTEPS = 2;                   // k = 32 channels per MFMA
constexpr int RS = 4;                       // tile rows per wave
constexpr int PB = 4, MB = 4;               // 16-pixel rows x 16-channel blocks per wave

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

}  // namespace

// CINW = words per input pixel (= F / 2).  ABL: timing-only ablation mask as in conv3x3_body.hip
// (1 no stores, 2 no residual loads, 4 no weight stream, 8 no input stream, 16 no barriers).
// PRE: accumulator pairs (0-2) whose residual values are fetched under the item's last step.
// ROLLW: register diet for PRE = 2: no second register set for the weight fragments (fragment mb of the next k-step
// is read into the registers of the current one right after its four MFMAs have issued, 12 MFMAs before its first
// use) and the input offsets recomputed per round (conv3x3_dma.h, LAZY_VOFF).
template <int CINW, int COUT, int EPI, int ABL, int PRE, bool ROLLW>
__global__ __launch_bounds__(THREADS, 2) void conv3x3_body16_kernel(const ConvParams p, const int n_items) {
  constexpr int NCC = CINW / KC;
  constexpr int NS = COUT / NT;
  static_assert(NCC % 2 == 0, "input double buffer parity");
  constexpr int N_W = (ABL & 4) ? 0 : 2;                   // weight DMAs per wave and step

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const in_s = smem;                            // [2][8][QS][4 words]
  float* const w_s = smem + 2 * IN_BYTES / 4;          // [4][8 k-groups][128 rows][4 words]
  float* const bias_s = w_s + NWBUF * WCH;             // [COUT]: the epilogue must not queue a vector-memory load
                                                       // behind its own stores

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
  // ---- the two DMA streams (conv3x3_dma.h) ----
  dma::Stage<CINW, NS, ROLLW> st;
  st.init(p, in_s, w_s, lane, wave, lid, G, n_items);

  // ---- per-lane operand addresses (words) ----
  // B operand (pixels): lane -> pixel column l15 of a 16-pixel row, channel group 4*s + q4 of the chunk
  // A operand (weights): [k-group = 4*s + q4][row = wn*64 + 16*mb + l15][4 words]
  const int x_lane = (q4 * QS + (RS * wp) * kHalo + l15) * 4;
  const int w_lane = (q4 * NT + wn * 64 + l15) * 4;

  // ---- prologue: first item's input chunk 0, weight chunks 0-2 ----
  st.set_stage_item(lid);
  if constexpr (!(ABL & 8)) {
#pragma unroll
    for (int b = 0; b < IN_BLOCKS; ++b) st.issue_in(0, b, 0);
  }
  if constexpr (!(ABL & 4)) {
    st.issue_w();
    st.issue_w();
    st.issue_w();
  }
  if (tid < COUT) bias_s[tid] = p.bias[tid];
  wait_vmcnt<0>();
  __syncthreads();

  f32x4 w_cur[MB], x_cur[PB], w_nxt[MB], x_nxt[PB];
  int mf_slot = 0;
  auto read_frags = [&](f32x4 (&wf)[MB], f32x4 (&xf)[PB], const float* ib, const float* wb, int tap, int s) {
    const int dy = tap / 3, dx = tap - dy * 3;
    const float* wp_ = wb + w_lane + (4 * s * NT) * 4;
    const float* xp_ = ib + x_lane + (4 * s * QS + dy * kHalo + dx) * 4;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) wf[mb] = *reinterpret_cast<const f32x4*>(wp_ + mb * 16 * 4);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) xf[pb] = *reinterpret_cast<const f32x4*>(xp_ + pb * kHalo * 4);
  };
  read_frags(w_cur, x_cur, in_s, w_s, 0, 0);
  auto read_x = [&](f32x4 (&xf)[PB], const float* ib, int tap, int s) {
    const int dy = tap / 3, dx = tap - dy * 3;
    const float* xp_ = ib + x_lane + (4 * s * QS + dy * kHalo + dx) * 4;
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) xf[pb] = *reinterpret_cast<const f32x4*>(xp_ + pb * kHalo * 4);
  };
  auto read_w1 = [&](int mb, const float* wb, int s) -> f32x4 {
    return *reinterpret_cast<const f32x4*>(wb + w_lane + (4 * s * NT) * 4 + mb * 16 * 4);
  };

  // The item loop is ROTATED: an iteration first writes out the PREVIOUS item's accumulators, then runs this item's
  // 36 steps (one extra iteration writes the last item; the first one stores to out-of-range offsets, which the
  // buffer unit drops).  Every path into an item's first step then passes the same epilogue, so its vector-memory
  // operations could be counted by that step's wait; the first input chunk (cc = 0) is its own copy of the step
  // code for the same reason.  (Counting them measured no gain — see the wait below — but the shape is kept: it
  // costs nothing and keeps that option one template argument away.)
  f32x4 acc[MB][PB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) acc[mb][pb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // residual values of the first PRE accumulator pairs, fetched at the head of an item's last step
  constexpr int kPre = EPI == kEpiResidual && !(ABL & 2) ? PRE : 0;
  f32x4 resv[kPre ? kPre : 1][PB][2];
#pragma unroll
  for (int pr = 0; pr < (kPre ? kPre : 1); ++pr)
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) resv[pr][pb][0] = resv[pr][pb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  // geometry of one item's output: descriptor of its image, the lane's element offset, validity
  struct OutGeom { int img; unsigned lane_eoff; int ey; bool col_ok; int ch8; };
  auto out_geom = [&](int item, bool valid) -> OutGeom {
    const int tile = item / NS, slab = item - tile * NS;
    const int img = tile / tiles_per_img;
    const int trem = tile - img * tiles_per_img;
    const int tyi = trem / p.tiles_x;
    const int ty0 = tyi * kTile, tx0 = (trem - tyi * p.tiles_x) * kTile;
    const int ch8 = slab * NT + wn * 64 + 8 * q4;
    const int ex = tx0 + l15, ey = ty0 + RS * wp;
    return OutGeom{img, (unsigned)((ey * p.w + ex) * COUT + ch8), ey, valid && ex < p.w, ch8};
  };
  auto aux_desc = [&](int img) {
    return __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.aux) + (EPI == kEpiResidual ? (size_t)img * img_pix * COUT : 0), 0,
        EPI == kEpiResidual ? (unsigned)(img_pix * COUT * 4) : 0, 0x00020000);
  };
  // elements; 0x20000000 is out of range once scaled to bytes (x2, x4)
  auto elem_off = [&](const OutGeom& g, int pr, int pb) -> unsigned {
    return g.col_ok && g.ey + pb < p.h ? g.lane_eoff + (unsigned)(pb * p.w * COUT + pr * 32) : 0x20000000u;
  };

  for (int it = 0; it <= my_items; ++it) {
    // ---- epilogue of item it-1: lane = pixel (ex, ey + pb), 8 consecutive channels ch8 + 32*pr per pair ----
    {
      const OutGeom g = out_geom(it > 0 ? lid + (it - 1) * G : lid, it > 0);
      // one buffer descriptor per image; pixels outside a ragged tile get an out-of-range offset, not a branch
      constexpr unsigned OB = EPI == kEpiRelu ? 2u : 4u;          // bytes per element of p.out
      const auto aux_rsrc = aux_desc(g.img);
      const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<char*>(p.out) + (size_t)g.img * img_pix * COUT * OB, 0, (unsigned)(img_pix * COUT * OB), 0x00020000);
      const auto out2_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<char*>(p.out2) + (EPI == kEpiResidual ? (size_t)g.img * img_pix * COUT * 2 : 0), 0,
          EPI == kEpiResidual ? (unsigned)(img_pix * COUT * 2) : 0, 0x00020000);
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias_s + g.ch8 + 32 * pr);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias_s + g.ch8 + 32 * pr + 4);
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
          f32x4 v0 = acc[2 * pr][pb] + b0, v1 = acc[2 * pr + 1][pb] + b1;
          const unsigned eo = elem_off(g, pr, pb);
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
              if (pr < kPre) {
                r0 = resv[pr][pb][0];
                r1 = resv[pr][pb][1];
              } else {
                r0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(aux_rsrc, eo * 4u, 0, kAuxNt));
                r1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(aux_rsrc, eo * 4u + 16u, 0, kAuxNt));
              }
            }
            v0 = r0 + v0 * p.res_scale;       // -ffp-contract=off: two roundings, as keras
            v1 = r1 + v1 * p.res_scale;
            const u32x4 h = {pack_bf16(v0[0], v0[1]), pack_bf16(v0[2], v0[3]), pack_bf16(v1[0], v1[1]), pack_bf16(v1[2], v1[3])};
            if constexpr (!(ABL & 1)) {
              // immediate soffset only: see the store-data hazard note in conv3x3_body.hip
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v0), out_rsrc, eo * 4u, 0, kAuxNt);
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v1), out_rsrc, eo * 4u + 16u, 0, kAuxNt);
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
      (void)first_c;      // cc == 0 is its own copy of the step code (kept: hipcc schedules it around the epilogue)
      const float* const ib = in_s + (cc & 1) * (IN_BYTES / 4);
      const float* const ib_next = in_s + ((cc + 1) & 1) * (IN_BYTES / 4);
      // what is staged into ib_next during this cc: (this item, cc+1), or on the last cc the NEXT item's chunk 0
      // (on the very last item: its own chunk 0 again, which nobody reads)
      const bool last_cc = cc == NCC - 1;
      const int in_cc = last_cc ? 0 : cc + 1;
      if (last_cc && have_next_item) st.set_stage_item(item + G);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float* const wb = w_s + mf_slot * WCH;
        const int nx_slot = mf_slot == NWBUF - 1 ? 0 : mf_slot + 1;
        const float* const wb_nx = w_s + nx_slot * WCH;
        constexpr int kNoIn = (ABL & 8) ? 1 : 0;
        const int n_in = (tap < IN_BLOCKS && !kNoIn) ? 1 : 0;     // folds: tap is an unrolled constant

#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
          // fragments of the next k-step: same chunk, or k-step 0 of the next (tap, chunk)
          const float* const ib_n = s < KSTEPS - 1 || tap < 8 ? ib : ib_next;
          const float* const wb_n = s < KSTEPS - 1 ? wb : wb_nx;
          const int tap_n = s < KSTEPS - 1 ? tap : tap < 8 ? tap + 1 : 0;
          const int s_n = s < KSTEPS - 1 ? s + 1 : 0;
          if constexpr (ROLLW)
            read_x(x_nxt, ib_n, tap_n, s_n);
          else
            read_frags(w_nxt, x_nxt, ib_n, wb_n, tap_n, s_n);
          if (s == 0) {
            // this step's DMAs: input round first, then the weight chunk three steps ahead
            if (n_in) st.issue_in((cc + 1) & 1, tap < IN_BLOCKS ? tap : 0, in_cc);
            if constexpr (!(ABL & 4)) st.issue_w();
            if constexpr (kPre > 0) {
              // the item's last step: its residual values, younger than the step's DMAs, land under its 32 MFMAs
              // (the wait that ends the step then covers them too, which is when they are needed)
              if (tap == 8 && last_cc) {
                const OutGeom g = out_geom(item, true);
                const auto aux_rsrc = aux_desc(g.img);
#pragma unroll
                for (int pr = 0; pr < kPre; ++pr)
#pragma unroll
                  for (int pb = 0; pb < PB; ++pb) {
                    const unsigned eo = elem_off(g, pr, pb);
                    resv[pr][pb][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(aux_rsrc, eo * 4u, 0, kAuxNt));
                    resv[pr][pb][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(aux_rsrc, eo * 4u + 16u, 0, kAuxNt));
                  }
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
              acc[mb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w_cur[mb]),
                                                                    __builtin_bit_cast(bf16x8, x_cur[pb]),
                                                                    acc[mb][pb], 0, 0, 0);
            if constexpr (ROLLW) {
              __builtin_amdgcn_sched_barrier(0);
              w_cur[mb] = read_w1(mb, wb_n, s_n);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if constexpr (!ROLLW) {
#pragma unroll
            for (int q = 0; q < MB; ++q) w_cur[q] = w_nxt[q];
          }
#pragma unroll
          for (int q = 0; q < PB; ++q) x_cur[q] = x_nxt[q];
        }
        mf_slot = nx_slot;
        __builtin_amdgcn_sched_barrier(0);
        // retire the weight chunk issued at the start of the PREVIOUS step (and everything older): what this wave
        // issued after it is AT LEAST this step's input round and weight chunk.  (After an epilogue its loads and
        // stores are younger too; counting them — vmcnt(N_W + 1 + 8) for the ReLU form — measured no faster, so the
        // simple count stays: the stores of an item drain within its successor's first step.)
        if (n_in) wait_vmcnt<N_W + 1>(); else wait_vmcnt<N_W>();
        if constexpr (!(ABL & 16)) __syncthreads();
      }
    };
    do_cc(0, std::true_type{});
#pragma unroll 1
    for (int cc = 1; cc < NCC; ++cc) do_cc(cc, std::false_type{});
  }
  wait_vmcnt<0>();       // no DMA may still be writing this workgroup's LDS when it is released
}

template <int CINW, int COUT, int EPI, int ABL = 0, int PRE = 0, bool ROLLW = false>
static hipError_t launch_body16_one(const ConvParams& p, hipStream_t stream) {
  auto kern = conv3x3_body16_kernel<CINW, COUT, EPI, ABL, PRE, ROLLW>;
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

hipError_t launch_conv3x3_body16(const ConvParams& p, int feat, int epilogue, int sub, hipStream_t stream) {
  if (epilogue == kEpiResidual && !p.out2) return hipErrorInvalidValue;
  if (feat == 128)       // DSen2 width in bf16 (not a BASELINE config): the default structure only
    return epilogue == kEpiRelu ? launch_body16_one<64, 128, kEpiRelu, 0, 0, true>(p, stream)
                                : launch_body16_one<64, 128, kEpiResidual, 0, 2, true>(p, stream);
  if (feat != 256) return hipErrorInvalidValue;
  if (g_body_ablate != 0) {
#define DSEN2_ABL(M)                                                                     \
  if (g_body_ablate == M)                                                                \
    return epilogue == kEpiRelu ? launch_body16_one<128, 256, kEpiRelu, M>(p, stream)    \
                                : launch_body16_one<128, 256, kEpiResidual, M>(p, stream);
    DSEN2_ABL(1) DSEN2_ABL(3) DSEN2_ABL(4) DSEN2_ABL(8) DSEN2_ABL(12) DSEN2_ABL(15) DSEN2_ABL(16) DSEN2_ABL(31)
#undef DSEN2_ABL
    return hipErrorInvalidValue;
  }
  if (epilogue == kEpiRelu)      // default: weight fragments reloaded in place (+0.1-0.7 % in the network); sub 3 = double buffered
    return sub == 3 ? launch_body16_one<128, 256, kEpiRelu>(p, stream) : launch_body16_one<128, 256, kEpiRelu, 0, 0, true>(p, stream);
  // sub: 0 = the whole residual tile prefetched under the last step, with the register diet that makes it fit
  //      (default: VDSen2 bf16 bench 14.42-14.54 k vs 14.22-14.24 k patches/s for 1, 14.03-14.08 k for 2);
  //      1 = first accumulator pair prefetched; 2 = no prefetch
  if (sub == 1) return launch_body16_one<128, 256, kEpiResidual, 0, 1>(p, stream);
  if (sub == 2) return launch_body16_one<128, 256, kEpiResidual, 0, 0>(p, stream);
  return launch_body16_one<128, 256, kEpiResidual, 0, 2, true>(p, stream);
}

}  // namespace dsen2
