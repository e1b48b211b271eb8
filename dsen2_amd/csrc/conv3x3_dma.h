// conv3x3_dma.h — the LDS-DMA staging of conv3x3_body32.hip (fp32 body convolution; the bf16 kernel,
// conv3x3_body16w.hip, carries its own wide-tile version of the same scheme).
//
// Byte geometry (identical for both: a step is (tap, 128 bytes of input channels per pixel)):
//   input chunk  : [channel group q: 8][pixel slot: 336][16 B]  = 43,008 B, double buffered; slots 0..323 = the
//                  18x18 halo tile in row-major order, group q = bytes 16q..16q+15 of the pixel's 128-byte chunk
//   weight chunk : [k-group: 8][row: 128][16 B] = 16 KiB, verbatim the packed layout of dsen2_internal.h; 4-slot ring
//   bias         : [COUT] floats
//
// Movement: `buffer_load_dwordx4 ... offen lds` — 1 KiB per wave instruction, LDS address = M0 + 16*lane, a lane whose
// offset is out of the descriptor's range writes zeros (= the convolution's zero padding), an EXEC-masked lane
// writes nothing.  Issued from inline asm: through the builtin hipcc tracks the DMA in its waitcnt model and turns
// every counted lgkmcnt(N) of the fragment pipeline into lgkmcnt(0).  hipcc therefore knows neither about these
// vector-memory operations nor about the LDS they write (M0 is saved and restored inside every asm statement), and
// the kernels spell the synchronisation out:
//   * vmcnt retires in ISSUE ORDER, all vector-memory operations of a wave together (loads, stores, DMAs);
//   * weight chunk c is issued in step c-3 into ring slot c%4 (last read in step c-4, which ended with a barrier),
//     retired by every wave's `s_waitcnt vmcnt(N)` at the END of step c-2 — N = a lower bound of what that wave
//     issued after it — followed by that step's barrier, and first read by the fragment prefetch at the end of c-1;
//   * the six rounds of an input chunk are issued in taps 0-5 of the previous chunk's steps (wave q moves channel
//     group q of 64 halo pixels per round) and are older than the weight DMA awaited at the end of tap 6; the chunk
//     is first read at the end of tap 8;
//   * a wait that assumes FEWER younger operations than there are is stricter, never weaker — so is every wait
//     hipcc inserts for the loads it does track (it counts none of the DMAs);
//   * a workgroup must not end with a DMA in flight (its LDS may already belong to the next one): vmcnt(0) at exit.
#pragma once
#include "dsen2_internal.h"

namespace dsen2 {
namespace dma {

constexpr int KC = 32;                      // 32-bit words per pixel and step (32 fp32 or 64 bf16 channels)
constexpr int NT = 128;                     // output channels per item
constexpr int THREADS = 512;                // 8 waves
constexpr int QS = 336;                     // pixel slots per channel-group row (>= 324 halo pixels, = 0 mod 16)
constexpr int IN_BYTES = 8 * QS * 16;       // one input chunk buffer
constexpr int IN_BLOCKS = 6;                // DMA rounds per chunk: 64 pixels each (the last one 16: slots 320-335)
constexpr int WCH = KC * NT;                // words per weight chunk (16 KiB)
constexpr int NWBUF = 4;
constexpr size_t LDS_BYTES = (size_t)2 * IN_BYTES + (size_t)NWBUF * WCH * 4 + 256 * 4;
static_assert(QS >= kHaloPix && QS % 16 == 0 && 64 * (IN_BLOCKS - 1) + 16 == QS, "input chunk geometry");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ unsigned lds_address(const float* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const float*)p;
}

// Per-thread state of the two streams.  CINW = 32-bit words per input pixel; NS = output slabs per tile.
// LAZY_VOFF: do not keep the six per-lane input offsets in registers across the item but recompute the one a round
// needs from the staged tile's origin (a dozen VALU operations per step; for kernels that are out of VGPRs).
template <int CINW, int NS, bool LAZY_VOFF = false>
struct Stage {
  static constexpr int NCC = CINW / KC;
  static constexpr int NCHUNK = NCC * 9;

  const float* in;                  // NHWC activations, CINW words per pixel
  int h, w, tiles_x, tiles_per_img;
  size_t img_pix;
  int lane, wave, lid, G, n_items;
  unsigned lds_in, lds_w;
  unsigned in_voff[LAZY_VOFF ? 1 : IN_BLOCKS];   // byte offset of (halo pixel 64*b + lane, group `wave`) inside the image; out of range = zero
  int st_y0, st_x0;                 // origin of the staged tile (LAZY_VOFF)
  __amdgpu_buffer_rsrc_t in_rsrc, w_rsrc;
  unsigned w_voff;
  int wl_item, wl_chunk, st_slot;   // item / chunk of the next weight DMA, ring slot it goes to

  __device__ __forceinline__ void init(const ConvParams& p, const float* in_s, const float* w_s, int lane_, int wave_,
                                       int lid_, int G_, int n_items_) {
    in = p.in; h = p.h; w = p.w; tiles_x = p.tiles_x; tiles_per_img = p.tiles_x * p.tiles_y;
    img_pix = (size_t)p.h * p.w;
    lane = lane_; wave = wave_; lid = lid_; G = G_; n_items = n_items_;
    lds_in = lds_address(in_s);
    lds_w = lds_address(w_s);
    in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, 0, 0x00020000);
    w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpk), 0, (unsigned)(NS * NCHUNK * WCH * 4), 0x00020000);
    w_voff = lane * 16;
    wl_item = lid; wl_chunk = 0; st_slot = 0;
    st_y0 = st_x0 = 0;
  }

  __device__ __forceinline__ unsigned voff_of(int b, int ty0, int tx0) const {
    const int hp = 64 * b + lane;
    const int hy = hp / kHalo, hx = hp - hy * kHalo;
    const int gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
    const bool inb = hp < kHaloPix && (unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)w;
    return inb ? (unsigned)(((gy * w + gx) * CINW + wave * 4) * 4) : 0x80000000u;
  }

  // the tile whose input the following issue_in() calls fetch
  __device__ __forceinline__ void set_stage_item(int item) {
    const int tile = item / NS;
    const int img = tile / tiles_per_img;
    const int trem = tile - img * tiles_per_img;
    const int tyi = trem / tiles_x;
    const int ty0 = tyi * kTile, tx0 = (trem - tyi * tiles_x) * kTile;
    in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in) + (size_t)img * img_pix * CINW, 0,
                                                (unsigned)(img_pix * CINW * 4), 0x00020000);
    if constexpr (LAZY_VOFF) {
      st_y0 = ty0;
      st_x0 = tx0;
    } else {
#pragma unroll
      for (int b = 0; b < IN_BLOCKS; ++b) in_voff[b] = voff_of(b, ty0, tx0);
    }
  }

  // round b of input chunk cc into buffer `buf` (one wave instruction per wave)
  __device__ __forceinline__ void issue_in(int buf, int b, int cc) {
    const unsigned m0v = lds_in + buf * IN_BYTES + (wave * QS + 64 * b) * 16;
    const unsigned so = cc * (KC * 4);
    const unsigned voff = LAZY_VOFF ? voff_of(b, st_y0, st_x0) : in_voff[LAZY_VOFF ? 0 : b];
    // M0 is saved and restored inside the statement: hipcc reserves M0 and rejects it in a clobber list, so the
    // compiler's own uses of M0 (none today) must never see the DMA's value
    unsigned saved_m0;
    if (b < IN_BLOCKS - 1) {
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(saved_m0) : "s"(m0v), "v"(voff), "s"(in_rsrc), "s"(so) : "memory");
    } else if (lane < 16) {          // slots 320-335 only: the next group's row starts at 336
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(saved_m0) : "s"(m0v), "v"(voff), "s"(in_rsrc), "s"(so) : "memory");
    }
  }

  // the next weight chunk of the stream (16 wave instructions of 1 KiB, two per wave) into the next ring slot; the
  // stream runs over item boundaries and, past the last item, wraps to this workgroup's first one (harmless)
  __device__ __forceinline__ void issue_w() {
    const unsigned so = (unsigned)(((wl_item % NS) * NCHUNK + wl_chunk) * (WCH * 4) + wave * 1024);
    const unsigned l0 = lds_w + st_slot * (WCH * 4) + wave * 1024;
    unsigned saved_m0;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %4, %5 offen lds\n\t"
        "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %4, %6 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(saved_m0) : "s"(l0), "s"(l0 + 8192u), "v"(w_voff), "s"(w_rsrc), "s"(so), "s"(so + 8192u)
        : "memory");
    if (++wl_chunk == NCHUNK) {
      wl_chunk = 0;
      wl_item = wl_item + G < n_items ? wl_item + G : lid;
    }
    st_slot = st_slot == NWBUF - 1 ? 0 : st_slot + 1;
  }
};

}  // namespace dma
}  // namespace dsen2
