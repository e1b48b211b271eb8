// conv3x3_body.hip — the hot kernel: F->F 3x3 'same' convolution (F = 128 or 256) on NHWC fp32, as a
// PERSISTENT, software-pipelined workgroup per CU.  Same arithmetic, operand layout and epilogues as
// conv3x3_mfma.hip (which remains the kernel for the 10/12->F input layer, the F->6/2 output layer and
// the reference structure for A/B runs); what changes is everything around the MFMA stream:
//
//   * one workgroup (8 waves, 2 per SIMD) per CU walks a list of (16x16 tile, 128-channel slab) items, so
//     the weight stream, the input-tile stream and the MFMA stream never drain between tiles: the next
//     item's first input chunk and first weight chunks are already in LDS when the current item's last
//     MFMA issues, and the epilogue's stores retire behind the next item's MFMAs;
//   * weight chunks (one per (tap, 32 input channels), 16 KiB) go through a 3-deep LDS ring, fetched two
//     chunks ahead (global -> VGPR at the top of a step, VGPR -> LDS at its bottom), so the single
//     s_barrier per step never has a just-written buffer read right behind it;
//   * operand fragments are double buffered in registers: the ds_read_b128s of k-step s+1 (or of the next
//     chunk's first k-step, which the ring makes legal) are issued BEFORE the 16 MFMAs of k-step s, so
//     LDS latency and the barrier sit behind MFMAs that are already queued on the matrix pipe.
//
// One step = one (tap, 32-channel chunk) = 64 MFMAs per wave (4 k-steps x 4 accumulators x 4).
#include <type_traits>

#include "dsen2_internal.h"

namespace dsen2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// (A global_load_lds weight stream was tried and dropped: with a DMA in flight hipcc turns every counted
//  lgkmcnt(N) of the fragment pipeline into lgkmcnt(0), which costs more than the ds_writes it saves.)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float quad_perm(float v) {     // DPP quad_perm lane exchange (hazards padded by hipcc)
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// 4x4 transpose across the 4 lanes of a quad: in (lane i, register j) -> out (lane j, register i).
// Turns the MFMA result "lane = channel, register = pixel" into "lane = pixel, registers = 4 consecutive
// channels", i.e. 16 contiguous bytes of an NHWC pixel per lane.  16 VALU ops, no LDS.
__device__ __forceinline__ void quad_transpose(float& a0, float& a1, float& a2, float& a3, bool odd, bool hi) {
  float s_, r_;
  s_ = odd ? a0 : a1; r_ = quad_perm<0xB1>(s_); a0 = odd ? r_ : a0; a1 = odd ? a1 : r_;     // lanes i <-> i^1
  s_ = odd ? a2 : a3; r_ = quad_perm<0xB1>(s_); a2 = odd ? r_ : a2; a3 = odd ? a3 : r_;
  s_ = hi ? a0 : a2; r_ = quad_perm<0x4E>(s_); a0 = hi ? r_ : a0; a2 = hi ? a2 : r_;        // lanes i <-> i^2
  s_ = hi ? a1 : a3; r_ = quad_perm<0x4E>(s_); a1 = hi ? r_ : a1; a3 = hi ? a3 : r_;
}

// KC = input channels per step, NWAVES = waves per workgroup:
//   <32, 8>: one workgroup per CU, wave tile 64 ch x 64 px      <16, 4>: two workgroups per CU, 64 ch x 128 px
template <int KC_, int NWAVES_>
struct BodyCfg {
  static constexpr int KC = KC_;
  static constexpr int NWAVES = NWAVES_;
  static constexpr int NT = 128;                      // output channels per item
  static constexpr int THREADS = 64 * NWAVES;
  static constexpr int PSTR = KC + 4;                 // floats per halo pixel in LDS (odd multiple of 16 B)
  static constexpr int IN_FLOATS = kHaloPix * PSTR;
  static constexpr int WCH = KC * NT;                 // floats per weight chunk
  static constexpr int QPP = KC / 4;
  static constexpr int IN_PIECES = kHaloPix * QPP;    // 16-byte pieces per input chunk
  static constexpr int IN_ROUNDS = (IN_PIECES + THREADS - 1) / THREADS;
  static constexpr int W_ROUNDS = (WCH / 4) / THREADS;
  static constexpr int NWBUF = 3;
  static constexpr int KSTEPS = KC / 8;
  static constexpr int RS = 16 / (NWAVES / 2);        // tile rows per wave strip
  static constexpr int PB = RS / 2;                   // 32-pixel (2 x 16) blocks per wave
  static constexpr int WG_PER_CU = 8 / NWAVES;
  static constexpr size_t LDS_BYTES = (size_t)(2 * IN_FLOATS + NWBUF * WCH) * sizeof(float);
  static_assert(IN_ROUNDS <= 8, "input round r is written mid tap r and must be visible to the prefetch at the end of tap 8");
  static_assert((WCH / 4) % THREADS == 0, "weight chunk splits evenly over the workgroup");
  static_assert(LDS_BYTES * WG_PER_CU <= 160 * 1024, "LDS budget");
};

// ABL: timing-only ablation mask (results are WRONG when non-zero; tools/ablate_body_conv.py):
//   1 = no output stores, 2 = no residual loads, 4 = no weight stream, 8 = no input stream, 16 = no barriers
//
// BF16 = true: the same kernel byte for byte on the memory side — an LDS/global "word" then holds two bf16
// channels, so CIN counts 32-bit WORDS per input pixel (= channels / 2), a 16-byte piece is 8 channels, a step is
// (tap, 64 channels) — and v_mfma_f32_32x32x16_bf16 takes a whole 16-byte fragment per instruction (lane l holds
// k = 8*(l>>5) .. +7, exactly the 8 consecutive channels one ds_read_b128 returns).  fp32 accumulate; kEpiRelu
// writes bf16; kEpiResidual keeps the residual stream in fp32 (aux/out) and also writes its bf16 copy (out2),
// which is what the next block's first convolution reads.
template <int CIN, int COUT, int EPI, int KC, int NWAVES, int ABL = 0, bool LT = false, bool BF16 = false, bool STG = false>
__global__ __launch_bounds__(64 * NWAVES, 2) void conv3x3_body_kernel(const ConvParams p, const int n_items) {
  using B = BodyCfg<KC, NWAVES>;
  constexpr int NT = B::NT, THREADS = B::THREADS, PSTR = B::PSTR, IN_FLOATS = B::IN_FLOATS, WCH = B::WCH;
  constexpr int QPP = B::QPP, IN_PIECES = B::IN_PIECES, IN_ROUNDS = B::IN_ROUNDS, W_ROUNDS = B::W_ROUNDS;
  constexpr int NWBUF = B::NWBUF, KSTEPS = B::KSTEPS, RS = B::RS, PB = B::PB;
  constexpr bool kPrefetchRes = EPI == kEpiResidual && NWAVES == 8;   // registers allow it only at 64x64 per wave
  constexpr int NCC = CIN / KC;            // 4 or 8 (even: the input double buffer re-aligns every item)
  constexpr int NCHUNK = NCC * 9;
  constexpr int NS = COUT / NT;            // output slabs per tile
  static_assert(NCC % 2 == 0, "input double buffer parity");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const in_s = smem;                       // [2][324][PSTR]
  float* const w_s = smem + 2 * IN_FLOATS;        // [3][KC/4][NT][4]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1;                 // 64-channel half of the slab
  const int wp = wave >> 1;                // RS-row strip of the tile
  const int l31 = lane & 31;
  const int hsel = lane >> 5;

  // persistent schedule: logical workgroup id (XCD-contiguous), items lid, lid+G, lid+2G, ...
  const int G = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  if (lid >= n_items) return;
  const int my_items = (n_items - lid + G - 1) / G;
  // Start stagger: all workgroups run the same program on equal work, so left alone they stay in lockstep and
  // every CU's epilogue (a burst of stores at several TB/s chip-wide) falls in the same few microseconds while
  // the matrix pipes idle.  Delaying workgroup k by (k mod 4) quarter-items spreads the bursts under other
  // workgroups' MFMA phases.  p.stagger = delay quantum in units of s_sleep 127 (8128 cycles); 0 = off.
  for (int d = (bid >> 3 & 3) * p.stagger; d > 0; --d) __builtin_amdgcn_s_sleep(127);
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const size_t img_pix = (size_t)p.h * p.w;

  // ---- input staging state: geometry of the item whose input is being PREFETCHED ----
  int g_off[IN_ROUNDS];    // float offset of this thread's 16-byte piece inside the image, -1 = zero padding
  int s_off[IN_ROUNDS];    // float offset inside an LDS input buffer, -1 = no piece
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
    stage_img = p.in + (size_t)img * img_pix * CIN;
#pragma unroll
    for (int r = 0; r < IN_ROUNDS; ++r) {
      const int piece = r * THREADS + tid;
      const int hp = piece / QPP, qq = piece - hp * QPP;
      const int hy = hp / kHalo, hx = hp - hy * kHalo;
      const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
      const bool inb = piece < IN_PIECES && (unsigned)gy < (unsigned)p.h && (unsigned)gx < (unsigned)p.w;
      g_off[r] = inb ? (gy * p.w + gx) * CIN + qq * 4 : -1;
    }
  };
  // Branch-free loads: hipcc waits vmcnt(0) at the join of any branch around a load, which would drain
  // every in-flight fetch at the top of a step.  Padding pixels load a valid address; the zero is
  // selected in when the piece is written to LDS.
  auto load_in = [&](int r, int cc) -> f32x4 {
    return *reinterpret_cast<const f32x4*>(stage_img + (g_off[r] >= 0 ? g_off[r] : 0) + cc * KC);
  };
  auto store_in = [&](float* buf, int r, f32x4 t) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = g_off[r] >= 0 ? t[e] : 0.f;
    if (s_off[r] >= 0) *reinterpret_cast<f32x4*>(buf + s_off[r]) = v;
  };

  // ---- weight stream, register staged, across item boundaries ----
  // chunk c (global step index) is LOADED (global -> VGPR) just before the last k-step of step c-3, WRITTEN
  // (VGPR -> LDS ring slot c%3) in the middle of step c-2, first READ by the fragment prefetch at the end of
  // step c-1.  Past the last item the stream keeps fetching harmlessly (no runtime branch around a load).
  int wl_item = lid;           // load side: item / chunk to fetch next
  int wl_chunk = 0;
  int st_slot = 0;             // store side: ring slot the next written chunk goes to
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

  // ---- per-lane operand addresses (floats) ----
  // pixel fragment (MFMA A operand): lane -> pixel (row l31>>4, col l31&15) of a 2x16 block, lanes 32-63 take
  // channels +4;  weight fragment (MFMA B operand): [g = 2s + hsel][o = l31][4]
  const int x_lane = ((l31 >> 4) * kHalo + (l31 & 15)) * PSTR + 4 * hsel + (RS * wp) * kHalo * PSTR;
  const int w_lane = (hsel * NT + wn * 64 + l31) * 4;

  // ---- prologue: first item's input chunk 0, weight chunks 0 and 1; chunk 2 and input round 0 in flight ----
  f32x4 wr[W_ROUNDS];          // weight chunk in flight (loaded at the tail of a step, written mid next step)
  f32x4 ir;                    // input piece in flight (same cadence)
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
  }
  __syncthreads();

  f32x4 w_cur[2], x_cur[PB], w_nxt[2], x_nxt[PB];
  int mf_slot = 0;             // ring slot of the chunk the MFMA stream is on
  auto read_frags = [&](f32x4 (&wf)[2], f32x4 (&xf)[PB], const float* ib, const float* wb, int tap, int s) {
    const int dy = tap / 3, dx = tap - dy * 3;
    const float* wp_ = wb + w_lane + (2 * s * NT) * 4;
    const float* xp_ = ib + x_lane + (dy * kHalo + dx) * PSTR + 8 * s;
    wf[0] = *reinterpret_cast<const f32x4*>(wp_);
    wf[1] = *reinterpret_cast<const f32x4*>(wp_ + 32 * 4);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) xf[pb] = *reinterpret_cast<const f32x4*>(xp_ + pb * 2 * kHalo * PSTR);
  };
  read_frags(w_cur, x_cur, in_s, w_s, 0, 0);

  // STG: the item loop exists twice, selected per wave at the top level: waves 0-3 stage (LDS writes + next loads)
  // ahead of k-step 0, waves 4-7 ahead of k-step KSTEPS/2, so the two waves of a SIMD are half a step apart.
  auto run = [&](auto mid_c) {
  constexpr int MIDS = decltype(mid_c)::value;
  for (int it = 0; it < my_items; ++it) {
    const int item = lid + it * G;
    const bool have_next_item = it + 1 < my_items;

    // this item's output geometry.  D = X(32 px x k) * W(k x 32 ch): a lane owns output channel l31 of the
    // block, register r owns pixel (r&3) + 8*(r>>2) + 4*hsel of the 2x16 block, so every epilogue access is
    // two full 128-byte lines per wave instruction (32 consecutive channels of 2 pixels).
    const int tile = item / NS, slab = item - tile * NS;
    const int img = tile / tiles_per_img;
    const int trem = tile - img * tiles_per_img;
    const int tyi = trem / p.tiles_x;
    const int ty0 = tyi * kTile, tx0 = (trem - tyi * p.tiles_x) * kTile;
    const bool full_tile = ty0 + kTile <= p.h && tx0 + kTile <= p.w;
    // Epilogue geometry.  The MFMA result has lane = output channel (l31), register r = pixel
    // (r&3) + 8*(r>>2) + 4*hsel of a 2x16 block.  A 4x4 transpose inside each lane quad turns register quad
    // g = r>>2 into "lane = pixel (l&3) + 8*(g&1) + 4*hsel of row g>>1, 4 registers = channels 4*(l31>>2) .. +3":
    // 16 contiguous bytes of NHWC per lane, 8 full 128-byte lines per wave instruction, 16 instead of 64 memory
    // instructions per lane.  Addressing through buffer descriptors (one per image): byte offset = lane part
    // (VGPR, fixed for the item) + a UNIFORM per-(mb, pb, g) part (SGPR soffset); elements outside a ragged
    // tile get an out-of-range lane offset instead of a branch (loads return 0, stores are dropped).
    constexpr bool kOutBf16 = BF16 && EPI == kEpiRelu;          // element type of p.out
    constexpr unsigned OB = kOutBf16 ? 2u : 4u;
    const int chq = slab * NT + wn * 64 + 4 * (l31 >> 2);       // first of this lane's 4 channels (block mb adds 32)
    const bool q_odd = lane & 1, q_hi = lane & 2;
    const unsigned img_bytes = (unsigned)(img_pix * COUT * sizeof(float));
    const auto aux_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.aux) + (EPI == kEpiResidual ? (size_t)img * img_pix * COUT : 0), 0,
        EPI == kEpiResidual ? img_bytes : 0, 0x00020000);
    const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(p.out) + (size_t)img * img_pix * COUT * OB, 0, (unsigned)(img_pix * COUT * OB), 0x00020000);
    const auto out2_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(p.out2) + (BF16 && EPI == kEpiResidual ? (size_t)img * img_pix * COUT * 2 : 0), 0,
        BF16 && EPI == kEpiResidual ? (unsigned)(img_pix * COUT * 2) : 0, 0x00020000);
    // element offset of (row 0 of this wave's strip, column (l&3) + 4*hsel, channel chq); (mb, pb, g) adds epi_eoff
    const int ex = tx0 + (lane & 3) + 4 * hsel, ey = ty0 + RS * wp;
    const unsigned lane_eoff = (unsigned)((ey * p.w + ex) * COUT + chq);
    auto epi_eoff = [&](int mb, int pb, int g) -> int { return ((2 * pb + (g >> 1)) * p.w + 8 * (g & 1)) * COUT + mb * 32; };
    auto epi_ok = [&](int pb, int g) -> bool {
      return full_tile || (ey + 2 * pb + (g >> 1) < p.h && ex + 8 * (g & 1) < p.w);
    };
    // kEpiRelu keeps the un-transposed form (lane = channel l31, register r = pixel (r&3) + 8*(r>>2) + 4*hsel: two
    // full 128-byte lines per dword store instruction): with nothing to load, the 256 VALU ops of the transposes
    // cost more than the 48 memory instructions they save (measured: conv-A 1.09 -> 1.11 ms fp32, 0.29 -> 0.33 bf16).
    const unsigned lane_eoff1 = (unsigned)((ey * p.w + tx0 + 4 * hsel) * COUT + slab * NT + wn * 64 + l31);
    auto epi_eoff1 = [&](int mb, int pb, int r) -> int {
      return ((2 * pb + (r >> 3)) * p.w + (r & 3) + 8 * ((r >> 2) & 1)) * COUT + mb * 32;
    };
    auto epi_ok1 = [&](int pb, int r) -> bool {
      return full_tile || (ey + 2 * pb + (r >> 3) < p.h && tx0 + 4 * hsel + (r & 3) + 8 * ((r >> 2) & 1) < p.w);
    };

    f32x16 acc[2][PB];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mb][pb][e] = 0.f;
    f32x4 resv[2][kPrefetchRes ? PB : 1][4];    // residual tile (16 B per register quad), fetched under the last step's MFMAs

#pragma unroll 1
    for (int cc = 0; cc < NCC; ++cc) {
      const float* const ib = in_s + (cc & 1) * IN_FLOATS;
      float* const ib_next = in_s + ((cc + 1) & 1) * IN_FLOATS;
      const bool last_cc = cc == NCC - 1;
      // What is prefetched into ib_next during this cc: (this item, cc+1), or on the last cc the NEXT item's
      // chunk 0 (on the very last item: its own chunk 0 again, which nobody reads).
      const int in_cc = last_cc ? 0 : cc + 1;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float* const wb = w_s + mf_slot * WCH;
        const int nx_slot = mf_slot == NWBUF - 1 ? 0 : mf_slot + 1;
        const float* const wb_nx = w_s + nx_slot * WCH;

#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
          // prefetch the fragments of the next k-step (same chunk, or step 0 of the next (tap, chunk))
          if (s < KSTEPS - 1) {
            read_frags(w_nxt, x_nxt, ib, wb, tap, s + 1);
          } else if (tap < 8) {
            read_frags(w_nxt, x_nxt, ib, wb_nx, tap + 1, 0);
          } else {
            // next chunk: other input buffer (next cc, or the next item's chunk 0 which also lives there)
            read_frags(w_nxt, x_nxt, ib_next, wb_nx, 0, 0);
          }
          if (s == MIDS) {
            // mid-step: the pieces in flight go to LDS here, so the end of the step is only the barrier
            // (nothing freshly written sits right behind it)
            if constexpr (!(ABL & 4)) store_w(wr);
            if constexpr (!(ABL & 8))
              if (tap < IN_ROUNDS) store_in(ib_next, tap < IN_ROUNDS ? tap : 0, ir);
          }
          if (LT ? s == KSTEPS - 1 : s == MIDS) {
            // issue the global loads of the pieces the NEXT step writes (LT: before this step's last 16 MFMAs;
            // otherwise right after this step's own LDS writes, which frees the staging registers first)
            if constexpr (!(ABL & 4)) load_w(wr);
            if constexpr (!(ABL & 8)) {
              if (tap + 1 < IN_ROUNDS) {
                ir = load_in(tap + 1 < IN_ROUNDS ? tap + 1 : 0, in_cc);
              } else if (tap == 8) {
                // round 0 of what the NEXT cc prefetches: (item, cc+2), or the next item's chunk 0 / chunk 1
                const int nn = cc + 2;
                if (nn == NCC && have_next_item) set_stage_item(item + G);
                ir = load_in(0, nn < NCC ? nn : nn - NCC);
              }
            }
          }
          if constexpr (kPrefetchRes && !(ABL & 2)) {
            // residual tile: fetched in one burst at the head of the item's last step, so it lands under that
            // step's 64 MFMAs (spreading the 64 loads over k-steps measured slower: longer live ranges, spills)
            if (tap == 8 && last_cc && s == 0) {
#pragma unroll
              for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                  for (int g = 0; g < 4; ++g)
                    resv[mb][pb][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        aux_rsrc, epi_ok(pb, g) ? lane_eoff * 4u : 0x80000000u, epi_eoff(mb, pb, g) * 4, 0));
            }
          }
          // pin the order: everything above is issued ahead of this k-step's MFMAs (hipcc otherwise sinks the
          // reads next to their first use and the wave stalls on LDS latency every other k-step)
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < (BF16 ? 1 : 4); ++j)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
              for (int pb = 0; pb < PB; ++pb) {
                if constexpr (BF16)
                  acc[mb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x_cur[pb]),
                                                                        __builtin_bit_cast(bf16x8, w_cur[mb]),
                                                                        acc[mb][pb], 0, 0, 0);
                else
                  acc[mb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(x_cur[pb][j], w_cur[mb][j], acc[mb][pb], 0, 0, 0);
              }
#pragma unroll
          for (int q = 0; q < 2; ++q) w_cur[q] = w_nxt[q];
#pragma unroll
          for (int q = 0; q < PB; ++q) x_cur[q] = x_nxt[q];
        }
        mf_slot = nx_slot;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 16)) __syncthreads();
      }
    }

    // ---- epilogue of this item (its stores retire behind the next item's MFMAs) ----
    if constexpr (EPI == kEpiRelu) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const float bias1 = p.bias[slab * NT + wn * 64 + l31 + mb * 32];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = fmaxf(acc[mb][pb][r] + bias1, 0.f);
            if constexpr (!(ABL & 1)) {
              if constexpr (kOutBf16)
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)v), out_rsrc,
                                                      epi_ok1(pb, r) ? lane_eoff1 * 2u : 0x80000000u,
                                                      epi_eoff1(mb, pb, r) * 2, 0);
              else
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), out_rsrc,
                                                      epi_ok1(pb, r) ? lane_eoff1 * 4u : 0x80000000u,
                                                      epi_eoff1(mb, pb, r) * 4, 0);
            } else {
              asm volatile("" ::"v"(v));                       // keep the accumulators live without storing
            }
          }
        }
      }
    } else {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias + chq + mb * 32);
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float a0 = acc[mb][pb][4 * g], a1 = acc[mb][pb][4 * g + 1], a2 = acc[mb][pb][4 * g + 2], a3 = acc[mb][pb][4 * g + 3];
          quad_transpose(a0, a1, a2, a3, q_odd, q_hi);
          f32x4 v = {a0 + bias[0], a1 + bias[1], a2 + bias[2], a3 + bias[3]};
          if constexpr (EPI == kEpiRelu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          } else {
            f32x4 rr = {1.f, 1.f, 1.f, 1.f};
            if constexpr (!(ABL & 2)) {
              if constexpr (kPrefetchRes)
                rr = resv[mb][pb][g];
              else   // two workgroups per CU: the other one's MFMAs cover this latency
                rr = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                   aux_rsrc, epi_ok(pb, g) ? lane_eoff * 4u : 0x80000000u,
                                                   epi_eoff(mb, pb, g) * 4, 0));
            }
            v = rr + v * p.res_scale;       // -ffp-contract=off: two roundings, as keras
          }
          const unsigned vo = epi_ok(pb, g) ? lane_eoff : 0x20000000u;      // elements; OOB once scaled to bytes
          if constexpr (!(ABL & 1)) {
            if constexpr (kOutBf16) {
              const u32x2 h = {(unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[0]) |
                                   ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[1]) << 16),
                               (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[2]) |
                                   ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[3]) << 16)};
              __builtin_amdgcn_raw_buffer_store_b64(h, out_rsrc, vo * 2u, epi_eoff(mb, pb, g) * 2, 0);
            } else {
              // soffset must NOT be a register here: gfx950 reads the 128-bit store data late (the last quad of
              // each 16-lane row last), and hipcc (ROCm 7.2) only pads the "store data overwritten too early"
              // hazard when soffset is an immediate — with an SGPR soffset the next VALU write of v's first
              // register (e.g. the next item's accumulator init) corrupted lanes 12-15 / 28-31 on some schedules.
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rsrc,
                                                     vo * 4u + (unsigned)(epi_eoff(mb, pb, g) * 4), 0, 0);
              if constexpr (BF16) {    // bf16 copy of the new residual stream for the next block's conv-A
                const u32x2 h = {(unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[0]) |
                                     ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[1]) << 16),
                                 (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[2]) |
                                     ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[3]) << 16)};
                __builtin_amdgcn_raw_buffer_store_b64(h, out2_rsrc, vo * 2u, epi_eoff(mb, pb, g) * 2, 0);
              }
            }
          } else {
            asm volatile("" ::"v"(v));                         // keep the accumulators live without storing
          }
        }
      }
    }
    }   // transposed (residual) epilogue
  }
  };   // run
  if constexpr (STG) {
    if (wave < NWAVES / 2)
      run(std::integral_constant<int, 0>{});
    else
      run(std::integral_constant<int, KSTEPS / 2>{});
  } else {
    run(std::integral_constant<int, KSTEPS / 2 - 1>{});
  }
}

int g_body_ablate = 0;
int g_body_stagger = 0;

template <int CIN, int COUT, int EPI, int KC = 32, int NWAVES = 8, int ABL = 0, bool LT = false, bool BF16 = false, bool STG = false>
static hipError_t launch_body_one(const ConvParams& p, hipStream_t stream) {
  using B = BodyCfg<KC, NWAVES>;
  auto kern = conv3x3_body_kernel<CIN, COUT, EPI, KC, NWAVES, ABL, LT, BF16, STG>;
  static bool attr_set[64] = {};
  static int cus[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)B::LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  const long long items = (long long)p.n * p.tiles_x * p.tiles_y * (COUT / B::NT);
  if (items <= 0 || items > 0x7fffffffLL) return hipErrorInvalidValue;
  const long long slots = (long long)cus[dev] * B::WG_PER_CU;     // persistent workgroups: every one resident
  const int grid = (int)(items < slots ? items : slots);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(B::THREADS), B::LDS_BYTES, stream, p, (int)items);
  return hipGetLastError();
}

hipError_t launch_conv3x3_body(const ConvParams& p, int feat, int epilogue, int variant, hipStream_t stream) {
  if (variant == 9 && feat == 128) {     // A/B: wave-group stagger of the staging slot
    return epilogue == kEpiRelu ? launch_body_one<128, 128, kEpiRelu, 32, 8, 0, false, false, true>(p, stream)
                                : launch_body_one<128, 128, kEpiResidual, 32, 8, 0, false, false, true>(p, stream);
  }
  if (feat == 128 && epilogue == kEpiRelu) return launch_body_one<128, 128, kEpiRelu>(p, stream);
  if (feat == 128 && epilogue == kEpiResidual) return launch_body_one<128, 128, kEpiResidual>(p, stream);
  if (feat == 256 && epilogue == kEpiRelu) return launch_body_one<256, 256, kEpiRelu>(p, stream);
  if (feat == 256 && epilogue == kEpiResidual) return launch_body_one<256, 256, kEpiResidual>(p, stream);
  return hipErrorInvalidValue;
}

// bf16 operands, fp32 accumulate: F -> F with F = 128 or 256 (CIN template argument = F/2 words per pixel)
hipError_t launch_conv3x3_body_bf16(const ConvParams& p, int feat, int epilogue, int variant, hipStream_t stream) {
  if (epilogue == kEpiResidual && !p.out2) return hipErrorInvalidValue;
  if (variant >= 4 && variant <= 7)      // 16x16x32 MFMA form fed by LDS-DMA (4-7 = its sub-variants 0-3)
    return launch_conv3x3_body16(p, feat, epilogue, variant - 4, stream);
  // default (2): conv-B on the deferred-epilogue kernel, conv-A on the wave-group-staggered persistent kernel
  if (variant == 2 && feat == 256 && g_body_ablate == 0 && epilogue == kEpiResidual && bodyd_supports(p, 256))
    return launch_conv3x3_bodyd(p, 256, epilogue, true, stream);
  if (variant == 2 && feat == 256 && g_body_ablate == 0 && epilogue == kEpiRelu)
    return launch_body_one<128, 256, kEpiRelu, 32, 8, 0, false, true, true>(p, stream);
  if (variant == 3 && feat == 256 && bodyd_supports(p, 256))      // A/B: deferred kernel for both epilogues
    return launch_conv3x3_bodyd(p, 256, epilogue, true, stream);
  if (feat == 256 && epilogue == kEpiRelu) return launch_body_one<128, 256, kEpiRelu, 32, 8, 0, false, true>(p, stream);
  if (feat == 256 && epilogue == kEpiResidual) return launch_body_one<128, 256, kEpiResidual, 32, 8, 0, false, true>(p, stream);
  if (feat == 128 && epilogue == kEpiRelu) return launch_body_one<64, 128, kEpiRelu, 32, 8, 0, false, true>(p, stream);
  if (feat == 128 && epilogue == kEpiResidual) return launch_body_one<64, 128, kEpiResidual, 32, 8, 0, false, true>(p, stream);
  return hipErrorInvalidValue;
}

}  // namespace dsen2
