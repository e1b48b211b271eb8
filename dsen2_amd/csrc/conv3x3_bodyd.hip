// conv3x3_bodyd.hip — the persistent body convolution with a DEFERRED epilogue.
//
// Same arithmetic, LDS layout, weight ring and fragment pipeline as conv3x3_body.hip; what changes is when an item's
// results leave the CU.  In conv3x3_body.hip every item ends with its epilogue (bias, ReLU or residual add, stores)
// while the matrix pipes idle: 2-5 % of an fp32 item, 25 % of a bf16 conv-B item whose stores run at HBM speed.
// Here the finished accumulators are copied to a second register set ("held") and the epilogue of item i is
// executed in 16 pieces inside the first 17 steps of item i+1, one piece per step, right behind that step's LDS
// writes: its residual load is issued one step before it is needed, its stores drain under the following MFMAs.
// The last item of a workgroup is flushed after its loop.
//
// Everything is compile-time indexed: the (channel chunk, tap) loops are fully unrolled (4 x 9 steps), so this
// kernel exists for 128 input WORDS per pixel only (fp32 F=128, bf16 F=256); other shapes use conv3x3_body.hip.
// No branch surrounds a load (hipcc would wait vmcnt(0) at the join): the first item's "held" epilogue runs
// with out-of-range buffer offsets instead of being skipped.
#include "dsen2_internal.h"

namespace dsen2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace bodyd {
constexpr int KC = 32, NT = 128, THREADS = 512, CINW = 128, NCC = CINW / KC, NCHUNK = NCC * 9;
constexpr int PSTR = KC + 4, IN_FLOATS = kHaloPix * PSTR, WCH = KC * NT, QPP = KC / 4;
constexpr int IN_PIECES = kHaloPix * QPP, IN_ROUNDS = (IN_PIECES + THREADS - 1) / THREADS, W_ROUNDS = (WCH / 4) / THREADS;
constexpr int NWBUF = 3, KSTEPS = KC / 8, NPIECE = 16;
constexpr unsigned kOob = 0xFFFFFF00u;      // byte offset beyond any buffer this kernel accepts (< 2^32 - 256 bytes)
constexpr size_t LDS_BYTES = (size_t)(2 * IN_FLOATS + NWBUF * WCH) * sizeof(float);
static_assert(IN_ROUNDS <= 8 && NCHUNK > NPIECE + 1, "pipeline depths");

template <int CTRL>
__device__ __forceinline__ float quad_perm(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// 4x4 transpose across the 4 lanes of a quad (see conv3x3_body.hip)
__device__ __forceinline__ void quad_transpose(float& a0, float& a1, float& a2, float& a3, bool odd, bool hi) {
  float s_, r_;
  s_ = odd ? a0 : a1; r_ = quad_perm<0xB1>(s_); a0 = odd ? r_ : a0; a1 = odd ? a1 : r_;
  s_ = odd ? a2 : a3; r_ = quad_perm<0xB1>(s_); a2 = odd ? r_ : a2; a3 = odd ? a3 : r_;
  s_ = hi ? a0 : a2; r_ = quad_perm<0x4E>(s_); a0 = hi ? r_ : a0; a2 = hi ? a2 : r_;
  s_ = hi ? a1 : a3; r_ = quad_perm<0x4E>(s_); a1 = hi ? r_ : a1; a3 = hi ? a3 : r_;
}
__device__ __forceinline__ u32x2 pack_bf16x4(const f32x4& v) {
  return u32x2{(unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[0]) |
                   ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[1]) << 16),
               (unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[2]) |
                   ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)v[3]) << 16)};
}
}  // namespace bodyd

template <int COUT, int EPI, bool BF16>
__global__ __launch_bounds__(bodyd::THREADS, 2) void conv3x3_bodyd_kernel(const ConvParams p, const int n_items) {
  using namespace bodyd;
  constexpr int CIN = CINW;
  constexpr int NS = COUT / NT;
  constexpr bool kOutBf16 = BF16 && EPI == kEpiRelu;
  constexpr unsigned OB = kOutBf16 ? 2u : 4u;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const in_s = smem;                       // [2][324][PSTR]
  float* const w_s = smem + 2 * IN_FLOATS;        // [3][KC/4][NT][4]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wp = wave >> 1;
  const int l31 = lane & 31, hsel = lane >> 5;
  const bool q_odd = lane & 1, q_hi = lane & 2;

  const int G = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  if (lid >= n_items) return;
  const int my_items = (n_items - lid + G - 1) / G;
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const size_t img_pix = (size_t)p.h * p.w;

  // ---- input staging (identical to conv3x3_body.hip) ----
  int g_off[IN_ROUNDS], s_off[IN_ROUNDS];
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
  auto load_in = [&](int r, int cc) -> f32x4 {
    return *reinterpret_cast<const f32x4*>(stage_img + (g_off[r] >= 0 ? g_off[r] : 0) + cc * KC);
  };
  auto store_in = [&](float* buf, int r, f32x4 t) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = g_off[r] >= 0 ? t[e] : 0.f;
    if (s_off[r] >= 0) *reinterpret_cast<f32x4*>(buf + s_off[r]) = v;
  };

  // ---- weight stream (identical to conv3x3_body.hip) ----
  int wl_item = lid, wl_chunk = 0, st_slot = 0;
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

  const int x_lane = ((l31 >> 4) * kHalo + (l31 & 15)) * PSTR + 4 * hsel + (4 * wp) * kHalo * PSTR;
  const int w_lane = (hsel * NT + wn * 64 + l31) * 4;

  // ---- epilogue state: whole-tensor buffer descriptors (the launcher guarantees every tensor < 4 GiB) ----
  const unsigned total_elems = (unsigned)(p.n * img_pix * COUT);
  const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, total_elems * OB, 0x00020000);
  const auto aux_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.aux), 0,
                                                          EPI == kEpiResidual ? total_elems * 4u : 0u, 0x00020000);
  const auto out2_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out2, 0, BF16 && EPI == kEpiResidual ? total_elems * 2u : 0u,
                                                           0x00020000);
  f32x16 held[2][2];             // finished accumulators of the previous item
  // geometry of the held item: element offset of (image, row 0 of this wave's strip, this lane's column, this lane's
  // first channel); for kEpiRelu the lane is a channel (l31) and the column part is 4*hsel, for kEpiResidual the
  // lane is pixel (l&3) + 4*hsel of a quad and owns channels 4*(l31>>2) .. +3 after the transpose
  unsigned held_eoff = 0;
  int held_ch = 0;               // first channel of the lane (bias index); block mb adds 32
  int held_ey = 0, held_ex = 0;  // image row of the strip / column of the lane, for ragged-tile masking
  bool held_valid = false, held_full = false;
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
      for (int e = 0; e < 16; ++e) held[mb][pb][e] = 0.f;

  f32x4 pend_res = {0.f, 0.f, 0.f, 0.f};      // residual quad of the piece in flight (kEpiResidual)
  f32x4 pend_bias = {0.f, 0.f, 0.f, 0.f};     // its bias quad (kEpiResidual) / .x = bias of block mb (kEpiRelu)

  // piece q of the held item = register quad g = q&3 of block (mb = q>>3, pb = (q>>2)&1)
  auto piece_off = [&](int q) -> int {        // uniform element offset of the piece relative to held_eoff
    const int mb = q >> 3, pb = (q >> 2) & 1, g = q & 3;
    if constexpr (EPI == kEpiRelu)            // registers 4g..4g+3 = pixels (0..3) + 8*(g&1) of row g>>1: base of r = 4g
      return ((2 * pb + (g >> 1)) * p.w + 8 * (g & 1)) * COUT + mb * 32;
    else
      return ((2 * pb + (g >> 1)) * p.w + 8 * (g & 1)) * COUT + mb * 32;
  };
  auto piece_ok = [&](int q, int dx) -> bool {  // dx: extra column inside the piece (kEpiRelu: register r&3)
    const int pb = (q >> 2) & 1, g = q & 3;
    return held_valid && (held_full || (held_ey + 2 * pb + (g >> 1) < p.h && held_ex + 8 * (g & 1) + dx < p.w));
  };
  auto issue_piece = [&](int q) {
    const int mb = q >> 3;
    if constexpr (EPI == kEpiResidual) {
      const unsigned vo = piece_ok(q, 0) ? (held_eoff + (unsigned)piece_off(q)) * 4u : kOob;
      pend_res = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(aux_rsrc, vo, 0, 0));
      pend_bias = *reinterpret_cast<const f32x4*>(p.bias + held_ch + mb * 32);
    } else {
      pend_bias[0] = p.bias[held_ch + mb * 32];
    }
  };
  auto finish_piece = [&](int q) {
    const int mb = q >> 3, pb = (q >> 2) & 1, g = q & 3;
    float a0 = held[mb][pb][4 * g], a1 = held[mb][pb][4 * g + 1], a2 = held[mb][pb][4 * g + 2], a3 = held[mb][pb][4 * g + 3];
    if constexpr (EPI == kEpiRelu) {
      const float b = pend_bias[0];
      const float v[4] = {fmaxf(a0 + b, 0.f), fmaxf(a1 + b, 0.f), fmaxf(a2 + b, 0.f), fmaxf(a3 + b, 0.f)};
#pragma unroll
      for (int e = 0; e < 4; ++e) {             // register r = 4g + e = pixel column +e: element offset + e*COUT
        const unsigned eo = held_eoff + (unsigned)(piece_off(q) + e * COUT);
        if constexpr (kOutBf16)
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)v[e]), out_rsrc,
                                                piece_ok(q, e) ? eo * 2u : kOob, 0, 0);
        else
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), out_rsrc,
                                                piece_ok(q, e) ? eo * 4u : kOob, 0, 0);
      }
    } else {
      quad_transpose(a0, a1, a2, a3, q_odd, q_hi);
      f32x4 v = {a0 + pend_bias[0], a1 + pend_bias[1], a2 + pend_bias[2], a3 + pend_bias[3]};
      v = pend_res + v * p.res_scale;           // -ffp-contract=off: two roundings, as keras
      const bool ok = piece_ok(q, 0);
      const unsigned eo = held_eoff + (unsigned)piece_off(q);
      // immediate soffset: see experiments/README.md (gfx950 store-data hazard with a register soffset)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), out_rsrc, ok ? eo * 4u : kOob, 0, 0);
      if constexpr (BF16) __builtin_amdgcn_raw_buffer_store_b64(pack_bf16x4(v), out2_rsrc, ok ? eo * 2u : kOob, 0, 0);
    }
  };

  // ---- prologue ----
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
  }
  __syncthreads();

  f32x4 w_cur[2], x_cur[2], w_nxt[2], x_nxt[2];
  int mf_slot = 0;
  auto read_frags = [&](f32x4 (&wf)[2], f32x4 (&xf)[2], const float* ib, const float* wb, int tap, int s) {
    const int dy = tap / 3, dx = tap - dy * 3;
    const float* wp_ = wb + w_lane + (2 * s * NT) * 4;
    const float* xp_ = ib + x_lane + (dy * kHalo + dx) * PSTR + 8 * s;
    wf[0] = *reinterpret_cast<const f32x4*>(wp_);
    wf[1] = *reinterpret_cast<const f32x4*>(wp_ + 32 * 4);
    xf[0] = *reinterpret_cast<const f32x4*>(xp_);
    xf[1] = *reinterpret_cast<const f32x4*>(xp_ + 2 * kHalo * PSTR);
  };
  read_frags(w_cur, x_cur, in_s, w_s, 0, 0);

  for (int it = 0; it < my_items; ++it) {
    const int item = lid + it * G;
    const bool have_next_item = it + 1 < my_items;

    f32x16 acc[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mb][pb][e] = 0.f;

    // An item is exactly NCHUNK = 36 steps = 0 mod 3 ring slots, so with the steps fully unrolled the stream
    // positions have the same value at the top of every item; left visible, hipcc hoists all 36 weight addresses
    // (and ring addresses) out of the item loop, spills them, and reloads one per step behind an s_waitcnt
    // vmcnt(0) that drains every load and store in flight.  Launder them so they are recomputed (one add) instead.
    asm volatile("" : "+s"(wl_chunk), "+s"(st_slot), "+s"(mf_slot));

#pragma unroll
    for (int cc = 0; cc < NCC; ++cc) {
      const float* const ib = in_s + (cc & 1) * IN_FLOATS;
      float* const ib_next = in_s + ((cc + 1) & 1) * IN_FLOATS;
      constexpr int kLast = NCC - 1;
      const int in_cc = cc == kLast ? 0 : cc + 1;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int st = cc * 9 + tap;                    // compile-time after unrolling
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
          if (s == KSTEPS / 2 - 1) {
            // mid-step: LDS writes of the pieces in flight, global loads of the next ones, one piece of the
            // PREVIOUS item's epilogue
            store_w(wr);
            if (tap < IN_ROUNDS) store_in(ib_next, tap < IN_ROUNDS ? tap : 0, ir);
            load_w(wr);
            if (tap + 1 < IN_ROUNDS) {
              ir = load_in(tap + 1 < IN_ROUNDS ? tap + 1 : 0, in_cc);
            } else if (tap == 8) {
              const int nn = cc + 2;
              if (nn == NCC && have_next_item) set_stage_item(item + G);
              ir = load_in(0, nn < NCC ? nn : nn - NCC);
            }
            if (st >= 1 && st <= NPIECE) finish_piece(st >= 1 && st <= NPIECE ? st - 1 : 0);
            if (st < NPIECE) issue_piece(st < NPIECE ? st : 0);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < (BF16 ? 1 : 4); ++j)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
              for (int pb = 0; pb < 2; ++pb) {
                if constexpr (BF16)
                  acc[mb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x_cur[pb]),
                                                                        __builtin_bit_cast(bf16x8, w_cur[mb]),
                                                                        acc[mb][pb], 0, 0, 0);
                else
                  acc[mb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(x_cur[pb][j], w_cur[mb][j], acc[mb][pb], 0, 0, 0);
              }
#pragma unroll
          for (int q = 0; q < 2; ++q) { w_cur[q] = w_nxt[q]; x_cur[q] = x_nxt[q]; }
        }
        mf_slot = nx_slot;
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
      }
    }

    // ---- hand the finished item over to the deferred epilogue ----
    {
      const int tile = item / NS, slab = item - tile * NS;
      const int img = tile / tiles_per_img;
      const int trem = tile - img * tiles_per_img;
      const int tyi = trem / p.tiles_x;
      const int ty0 = tyi * kTile, tx0 = (trem - tyi * p.tiles_x) * kTile;
      held_full = ty0 + kTile <= p.h && tx0 + kTile <= p.w;
      held_valid = true;
      held_ey = ty0 + 4 * wp;
      if constexpr (EPI == kEpiRelu) {
        held_ex = tx0 + 4 * hsel;
        held_ch = slab * NT + wn * 64 + l31;
      } else {
        held_ex = tx0 + (lane & 3) + 4 * hsel;
        held_ch = slab * NT + wn * 64 + 4 * (l31 >> 2);
      }
      held_eoff = (unsigned)((((size_t)img * p.h + held_ey) * p.w + held_ex) * COUT + held_ch);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) held[mb][pb] = acc[mb][pb];
    }
  }

  // ---- flush: the last item's epilogue ----
#pragma unroll
  for (int q = 0; q < NPIECE; ++q) {
    issue_piece(q);
    finish_piece(q);
  }
}

template <int COUT, int EPI, bool BF16>
static hipError_t launch_bodyd_one(const ConvParams& p, hipStream_t stream) {
  auto kern = conv3x3_bodyd_kernel<COUT, EPI, BF16>;
  static bool attr_set[64] = {};
  static int cus[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bodyd::LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    attr_set[dev] = true;
  }
  const long long items = (long long)p.n * p.tiles_x * p.tiles_y * (COUT / bodyd::NT);
  if (items <= 0 || items > 0x7fffffffLL) return hipErrorInvalidValue;
  const int grid = (int)(items < cus[dev] ? items : cus[dev]);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(bodyd::THREADS), bodyd::LDS_BYTES, stream, p, (int)items);
  return hipGetLastError();
}

// true when conv3x3_bodyd_kernel can run this problem: whole output tensor addressable with 32-bit byte offsets
bool bodyd_supports(const ConvParams& p, int cout) {
  const unsigned long long bytes = (unsigned long long)p.n * p.h * p.w * cout * 4ull;
  return bytes < 0xFFFFFF00ull;
}

hipError_t launch_conv3x3_bodyd(const ConvParams& p, int feat, int epilogue, bool bf16, hipStream_t stream) {
  if (!bf16 && feat == 128)
    return epilogue == kEpiRelu ? launch_bodyd_one<128, kEpiRelu, false>(p, stream)
                                : launch_bodyd_one<128, kEpiResidual, false>(p, stream);
  if (bf16 && feat == 256) {
    if (epilogue == kEpiResidual && !p.out2) return hipErrorInvalidValue;
    return epilogue == kEpiRelu ? launch_bodyd_one<256, kEpiRelu, true>(p, stream)
                                : launch_bodyd_one<256, kEpiResidual, true>(p, stream);
  }
  return hipErrorInvalidValue;
}

}  // namespace dsen2
