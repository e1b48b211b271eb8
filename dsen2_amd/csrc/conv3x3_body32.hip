// conv3x3_body32.hip — fp32 F->F 3x3 'same' convolution (F = 128 or 256): v_mfma_f32_32x32x2_f32 fed by LDS-DMA.
//
// The fp32 sibling of conv3x3_body16w.hip (the DMA streams and their synchronisation rules: conv3x3_dma.h; the
// byte geometry is identical: a step is (tap, 32 channels) = 128 B per pixel and a 16 KiB weight chunk):
//
//   * weight chunks and input chunks go global -> LDS by `buffer_load_dwordx4 ... lds` issued from inline asm and
//     retired by hand-counted `s_waitcnt vmcnt(N)` + the step barrier: no staging VGPRs, ds_writes or selects
//     (out-of-range lanes write the zero padding);
//   * the input chunk lives in LDS as [channel group q: 8][pixel slot: 336][16 B]; a wave's 32-pixel block is rows
//     (R, R + 8) x 16 columns of the tile — 8 halo rows = 144 slots = 0 mod 16 apart — so every ds_read_b128 of the
//     loop is bank-conflict free (an adjacent-row 2x16 block is 2-way);
//   * the WEIGHTS are the MFMA's A operand: D[row = output channel][col = pixel].  A lane owns one pixel and, per
//     accumulator, channels (r & 3) + 8*(r >> 2) + 4*(lane >> 5): register quad g = 16 contiguous bytes of NHWC,
//     the two half-waves together 32 contiguous bytes per pixel.  The epilogue is 16-byte loads/stores with no
//     cross-lane transposes (the X-as-A form needs 64 dword stores or 256 DPP moves per lane and item);
//   * default (DEFER): the finished accumulators are copied to a second register set — the DMA staging freed the
//     registers — and written out in 16 pieces inside the next item's first 8 steps, residual quads fetched one
//     step ahead; without DEFER the residual tile is fetched under the item's last step (PRE) and the epilogue's
//     stores drain under the next item's first step (3.5 us);
//   * waves 4-7 issue their DMAs half a step after waves 0-3 (STG).
//
// Per wave: 64 channels x 64 pixels = 2 x 2 accumulators of 32x32; one step = 4 k-steps of 16 MFMAs and
// 4 ds_read_b128.  Same arithmetic order per output element as the one-tile-per-workgroup kernel of conv3x3_mfma.hip:
// results are bit-identical (tests/test_gpu_conv.py).
#include <type_traits>

#include "conv3x3_dma.h"

namespace dsen2 {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

using dma::KC; using dma::NT; using dma::THREADS; using dma::QS; using dma::IN_BYTES; using dma::IN_BLOCKS;
using dma::WCH; using dma::NWBUF; using dma::LDS_BYTES; using dma::wait_vmcnt;
constexpr int KSTEPS = 4;                   // 8 channels per k-step: one ds_read_b128 feeds 4 MFMAs (k = 2 each)
constexpr int PB = 2, MB = 2;               // 32-pixel blocks x 32-channel blocks per wave
static_assert((8 * kHalo) % 16 == 0, "rows R and R+8 of a pixel block sit on the same banks");

}  // namespace

// ABL: timing-only ablation mask (1 no stores, 2 no residual loads, 4 no weight stream, 8 no input stream,
// 16 no barriers).  PRE: 32-channel blocks (0-2) whose residual values are fetched under the item's last step.
// STG: waves 4-7 issue their DMAs half a step after waves 0-3 (two copies of the item loop).
// DEFER: the epilogue of item i is cut into 16 pieces (one 16-byte register quad each) that run inside the first
// 8 steps of item i+1, out of a copy of the accumulators: loads and stores trickle out under MFMAs instead of in a
// burst between items (PRE is then unused: a piece's residual quad is fetched one step before the piece runs).
template <int CIN, int COUT, int EPI, int ABL, int PRE, bool STG, bool DEFER>
__global__ __launch_bounds__(THREADS, 2) void conv3x3_body32_kernel(const ConvParams p, const int n_items) {
  constexpr int NCC = CIN / KC;
  constexpr int NS = COUT / NT;
  static_assert(NCC % 2 == 0, "input double buffer parity");
  constexpr int N_W = (ABL & 4) ? 0 : 2;                   // weight DMAs per wave and step

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const in_s = smem;                            // [2][8][QS][4 words]
  float* const w_s = smem + 2 * IN_BYTES / 4;          // [4][8 k-groups][128 rows][4 words]
  float* const bias_s = w_s + NWBUF * WCH;             // [COUT]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1;                 // 64-channel half of the slab
  const int wp = wave >> 1;                // rows 2*wp, 2*wp + 1 and the same + 8
  const int l31 = lane & 31;
  const int hsel = lane >> 5;
  const int rh = l31 >> 4, c16 = l31 & 15;

  // persistent schedule: logical ids remapped so that each XCD (blockIdx % 8) walks a contiguous run of items
  const int G = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  if (lid >= n_items) return;
  const int my_items = (n_items - lid + G - 1) / G;
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const size_t img_pix = (size_t)p.h * p.w;

  // ---- the two DMA streams (conv3x3_dma.h) ----
  dma::Stage<CIN, NS> st;
  st.init(p, in_s, w_s, lane, wave, lid, G, n_items);

  // ---- per-lane operand addresses (words) ----
  // B operand (pixels): lane -> pixel (row 2*wp + pb + 8*rh, column c16), channel group 2*s + hsel of the chunk
  // A operand (weights): [k-group = 2*s + hsel][row = wn*64 + 32*mb + l31][4 words]
  const int x_lane = (hsel * QS + (2 * wp + 8 * rh) * kHalo + c16) * 4;
  const int w_lane = (hsel * NT + wn * 64 + l31) * 4;

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
    const float* wp_ = wb + w_lane + (2 * s * NT) * 4;
    const float* xp_ = ib + x_lane + (2 * s * QS + dy * kHalo + dx) * 4;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) wf[mb] = *reinterpret_cast<const f32x4*>(wp_ + mb * 32 * 4);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) xf[pb] = *reinterpret_cast<const f32x4*>(xp_ + pb * kHalo * 4);
  };
  read_frags(w_cur, x_cur, in_s, w_s, 0, 0);

  // geometry of one item's output: its image, the lane's element offset (row 2*wp + 8*rh, column c16, channel
  // base of the lane), validity of the lane's column
  struct OutGeom { int img; unsigned lane_eoff; int ey; bool col_ok; int chb; };
  auto out_geom = [&](int item, bool valid) -> OutGeom {
    const int tile = item / NS, slab = item - tile * NS;
    const int img = tile / tiles_per_img;
    const int trem = tile - img * tiles_per_img;
    const int tyi = trem / p.tiles_x;
    const int ty0 = tyi * kTile, tx0 = (trem - tyi * p.tiles_x) * kTile;
    const int chb = slab * NT + wn * 64 + 4 * hsel;
    const int ex = tx0 + c16, ey = ty0 + 2 * wp + 8 * rh;
    return OutGeom{img, (unsigned)((ey * p.w + ex) * COUT + chb), ey, valid && ex < p.w, chb};
  };
  auto aux_desc = [&](int img) {
    return __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.aux) + (EPI == kEpiResidual ? (size_t)img * img_pix * COUT : 0), 0,
        EPI == kEpiResidual ? (unsigned)(img_pix * COUT * 4) : 0, 0x00020000);
  };
  // byte offset of register quad g of accumulator (mb, pb); out of range for pixels outside a ragged tile
  auto byte_off = [&](const OutGeom& g_, int mb, int pb, int g) -> unsigned {
    return g_.col_ok && g_.ey + pb < p.h ? (g_.lane_eoff + (unsigned)(pb * p.w * COUT + mb * 32 + 8 * g)) * 4u : 0x80000000u;
  };

  auto run = [&](auto mid_c) {
  constexpr int MIDS = decltype(mid_c)::value;
  f32x16 acc[MB][PB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mb][pb][e] = 0.f;
  // residual values of the first PRE 32-channel blocks, fetched at the head of an item's last step
  constexpr int kPre = EPI == kEpiResidual && !(ABL & 2) && !DEFER ? PRE : 0;
  f32x4 resv[kPre ? kPre : 1][PB][4];
#pragma unroll
  for (int mb = 0; mb < (kPre ? kPre : 1); ++mb)
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int g = 0; g < 4; ++g) resv[mb][pb][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Rotated item loop: an iteration first takes over the PREVIOUS item's accumulators (writes them out, or with
  // DEFER copies them to `held`), then runs this item's steps (one extra iteration handles the last item; the
  // first one stores zeros to out-of-range offsets, which are dropped).
  f32x16 held[DEFER ? MB : 1][DEFER ? PB : 1];
  f32x4 rq[DEFER && EPI == kEpiResidual ? 4 : 1];      // residual quads in flight: pieces 2t, 2t+1 and 2t+2, 2t+3
  for (int it = 0; it <= my_items; ++it) {
    const OutGeom g_ = out_geom(it > 0 ? lid + (it - 1) * G : lid, it > 0);
    const auto aux_rsrc = aux_desc(g_.img);
    const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.out + (size_t)g_.img * img_pix * COUT, 0, (unsigned)(img_pix * COUT * 4), 0x00020000);
    // piece j = register quad g of accumulator (mb, pb): j = 8*mb + 4*pb + g, so that the two pieces of a step are
    // 64 contiguous, 64-byte aligned bytes of one pixel (with j = 8*mb + 2*g + pb — 32-byte fragments of two pixels,
    // their neighbours written microseconds later — WRITE_SIZE read 1.56 x the tensor)
    auto load_res = [&](int j) -> f32x4 {
      const int mb = j >> 3, pb = (j >> 2) & 1, g = j & 3;
      return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(aux_rsrc, byte_off(g_, mb, pb, g), 0, 0));
    };
    auto finish = [&](int j, const f32x16& a, f32x4 rr) {
      const int mb = j >> 3, pb = (j >> 2) & 1, g = j & 3;
      const f32x4 bias = *reinterpret_cast<const f32x4*>(bias_s + g_.chb + mb * 32 + 8 * g);
      f32x4 v = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
      v = v + bias;
      if constexpr (EPI == kEpiRelu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      } else {
        v = rr + v * p.res_scale;       // -ffp-contract=off: two roundings, as keras
      }
      if constexpr (!(ABL & 1))
        // soffset must NOT be a register here: gfx950 reads the 128-bit store data late (the last quad of each
        // 16-lane row last), and hipcc (ROCm 7.2) only pads the "store data overwritten too early" hazard when
        // soffset is an immediate — with an SGPR soffset the next VALU write of v's first register corrupted lanes
        // 12-15 / 28-31 on some schedules (experiments/README.md; tests/test_gpu_stress.py is the screen).
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rsrc, byte_off(g_, mb, pb, g), 0, 0);
      else
        asm volatile("" ::"v"(v));
    };
    constexpr bool kRes = EPI == kEpiResidual && !(ABL & 2);
    if constexpr (!DEFER) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int mb = j >> 3, pb = (j >> 2) & 1, g = j & 3;
        f32x4 rr = {1.f, 1.f, 1.f, 1.f};
        if constexpr (kRes) rr = mb < kPre ? resv[mb][pb][g] : load_res(j);
        finish(j, acc[mb][pb], rr);
      }
    } else {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) held[mb][pb] = acc[mb][pb];
      if (it == my_items) {       // nothing left to hide behind: write the last item out in one go
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          f32x4 rr = {1.f, 1.f, 1.f, 1.f};
          if constexpr (kRes) rr = load_res(j);
          finish(j, held[j >> 3][(j >> 2) & 1], rr);
        }
      } else if constexpr (kRes) {
        rq[0] = load_res(0);      // pieces 0 and 1 run in the first step
        rq[1] = load_res(1);
      }
    }
    if (it == my_items) break;
    const int item = lid + it * G;
    const bool have_next_item = it + 1 < my_items;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mb][pb][e] = 0.f;

    auto do_cc = [&](const int cc, auto first_c) {
      constexpr bool kFirst = decltype(first_c)::value;      // cc == 0 is its own copy of the step code
      (void)kFirst;
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
          if (s < KSTEPS - 1) {
            read_frags(w_nxt, x_nxt, ib, wb, tap, s + 1);
          } else if (tap < 8) {
            read_frags(w_nxt, x_nxt, ib, wb_nx, tap + 1, 0);
          } else {
            read_frags(w_nxt, x_nxt, ib_next, wb_nx, 0, 0);
          }
          if constexpr (DEFER) {
            // first input chunk of an item only (this copy of the code): step `tap` finishes pieces 2*tap and
            // 2*tap + 1 of the previous item; their residual quads were fetched one step earlier, BEFORE that
            // step's DMAs in issue order, so waiting for them never waits for a DMA younger than a step
            if (kFirst && tap < 8) {
              if (s == MIDS && kRes && tap < 7) {
                rq[2 * ((tap + 1) & 1)] = load_res(2 * tap + 2);
                rq[2 * ((tap + 1) & 1) + 1] = load_res(2 * tap + 3);
              }
            }
          }
          if (s == MIDS) {
            // this step's DMAs: input round first, then the weight chunk three steps ahead
            if (n_in) st.issue_in((cc + 1) & 1, tap < IN_BLOCKS ? tap : 0, in_cc);
            if constexpr (!(ABL & 4)) st.issue_w();
          }
          if constexpr (DEFER) {
            if (kFirst && tap < 8 && (s == 1 || s == 3)) {
              const int j = 2 * tap + (s >> 1);
              f32x4 rr = {1.f, 1.f, 1.f, 1.f};
              if constexpr (kRes) rr = rq[2 * (tap & 1) + (s >> 1)];
              finish(j, held[j >> 3][(j >> 2) & 1], rr);
            }
          }
          if constexpr (kPre > 0 && !DEFER) {
            // the item's last step: its residual values, younger than every DMA this wave waits for before the
            // epilogue, land under the step's 64 MFMAs
            if (tap == 8 && last_cc && s == (MIDS < KSTEPS - 1 ? MIDS + 1 : MIDS)) {
              const OutGeom g_ = out_geom(item, true);
              const auto aux_rsrc = aux_desc(g_.img);
#pragma unroll
              for (int mb = 0; mb < kPre; ++mb)
#pragma unroll
                for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                  for (int g = 0; g < 4; ++g)
                    resv[mb][pb][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                    aux_rsrc, byte_off(g_, mb, pb, g), 0, 0));
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
              for (int pb = 0; pb < PB; ++pb)
                acc[mb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w_cur[mb][j], x_cur[pb][j], acc[mb][pb], 0, 0, 0);
#pragma unroll
          for (int q = 0; q < MB; ++q) w_cur[q] = w_nxt[q];
#pragma unroll
          for (int q = 0; q < PB; ++q) x_cur[q] = x_nxt[q];
        }
        mf_slot = nx_slot;
        __builtin_amdgcn_sched_barrier(0);
        // retire the weight chunk issued in the PREVIOUS step (and everything older): what this wave issued after
        // it is AT LEAST this step's input round and weight chunk (anything more only makes the wait stricter)
        if (n_in) wait_vmcnt<N_W + 1>(); else wait_vmcnt<N_W>();
        if constexpr (!(ABL & 16)) __syncthreads();
      }
    };
    do_cc(0, std::true_type{});
#pragma unroll 1
    for (int cc = 1; cc < NCC; ++cc) do_cc(cc, std::false_type{});
  }
  };   // run
  if constexpr (STG) {
    if (wave < 4)
      run(std::integral_constant<int, 0>{});
    else
      run(std::integral_constant<int, KSTEPS / 2>{});
  } else {
    run(std::integral_constant<int, 0>{});
  }
  wait_vmcnt<0>();       // no DMA may still be writing this workgroup's LDS when it is released
}

template <int CIN, int COUT, int EPI, int ABL = 0, int PRE = 0, bool STG = false, bool DEFER = false>
static hipError_t launch_body32_one(const ConvParams& p, hipStream_t stream) {
  auto kern = conv3x3_body32_kernel<CIN, COUT, EPI, ABL, PRE, STG, DEFER>;
  static KernelOnce once;
  int cus = 0;
  hipError_t e = once.prepare(reinterpret_cast<const void*>(kern), LDS_BYTES, &cus);
  if (e != hipSuccess) return e;
  const long long items = (long long)p.n * p.tiles_x * p.tiles_y * (COUT / NT);
  if (items <= 0 || items > 0x7fffffffLL) return hipErrorInvalidValue;
  const int grid = (int)(items < cus ? items : cus);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), LDS_BYTES, stream, p, (int)items);
  return hipGetLastError();
}

// per-image buffer descriptors address bytes with 32 bits (and keep 0x80000000 out of range)
bool body32_supports(const ConvParams& p, int cout) { return (size_t)p.h * p.w * cout * 4 < 0x80000000ull; }

template <int F>
static hipError_t launch_body32_feat(const ConvParams& p, int epilogue, int sub, hipStream_t stream) {
  if (epilogue == kEpiRelu)
    return sub == 1   ? launch_body32_one<F, F, kEpiRelu, 0, 0, false>(p, stream)
           : sub == 0 ? launch_body32_one<F, F, kEpiRelu, 0, 0, true>(p, stream)
                      : launch_body32_one<F, F, kEpiRelu, 0, 0, true, true>(p, stream);
  if (sub == 2) return launch_body32_one<F, F, kEpiResidual, 0, 0, false, true>(p, stream);
  if (sub == 3) return launch_body32_one<F, F, kEpiResidual, 0, 0, true, true>(p, stream);
  return launch_body32_one<F, F, kEpiResidual, 0, 2, false>(p, stream);
}

// sub: 3 = deferred epilogue + wave-group stagger for both convolutions (default);
//      2 = the same with conv-B not staggered; 0 = no deferral: conv-A staggered, conv-B with the whole residual
//      tile prefetched under the last step; 1 = 0 without the stagger
hipError_t launch_conv3x3_body32(const ConvParams& p, int feat, int epilogue, int sub, int ablate, hipStream_t stream) {
  if (!body32_supports(p, feat)) return hipErrorInvalidValue;
#ifdef DSEN2_DIAG
  if (feat == 128 && ablate != 0) {
#define DSEN2_ABL(M)                                                                             \
  if (ablate == M)                                                                               \
    return epilogue == kEpiRelu ? launch_body32_one<128, 128, kEpiRelu, M>(p, stream)            \
                                : launch_body32_one<128, 128, kEpiResidual, M, 2>(p, stream);
    DSEN2_ABL(1) DSEN2_ABL(2) DSEN2_ABL(3) DSEN2_ABL(4) DSEN2_ABL(8) DSEN2_ABL(12) DSEN2_ABL(15) DSEN2_ABL(16) DSEN2_ABL(31)
#undef DSEN2_ABL
    return hipErrorInvalidValue;
  }
#endif
  if (ablate != 0) return hipErrorInvalidValue;     // timing-only ablations exist in the diagnostic build only
  if (feat == 128) return launch_body32_feat<128>(p, epilogue, sub, stream);
  if (feat == 256) return launch_body32_feat<256>(p, epilogue, sub, stream);
  return hipErrorInvalidValue;
}

}  // namespace dsen2
