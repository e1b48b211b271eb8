// conv3x3_body16w.hip — bf16-operand F->F 3x3 'same' convolution (F = 128 or 256) of the residual blocks
// (utils/DSen2Net.py:9-15 with precision = 1): v_mfma_f32_16x16x32_bf16 fed by LDS-DMA, WIDE pixel tile.
//
// Round-1's kernel (16x16 pixels x 128 channels per item) spent its time in the CU's vector-memory path, not
// in the matrix pipe (profiles/archive/r01_ablation.md): per 151 MFLOP item it pulled 576 KiB of weights + 162 KiB of
// input through L1 into LDS and, for conv-B, pushed 24 store instructions per lane back through the same path.
// This kernel changes the three ratios that set that traffic:
//
//   1. ITEM = 16 rows x 32 columns of pixels x 128 output channels.  A weight chunk now feeds 512 pixels instead
//      of 256: L2->LDS bytes per MFMA fall by 40 % (weights 288 KiB + input 153 KiB per 151 MFLOP).  A wave's tile
//      is 64 channels x (8 rows x 16 columns) = 4 x 8 accumulators.
//   2. STEP = (tap, 32 input channels) = one MFMA k-step: an 8 KiB weight chunk, so the weight ring holds 8 chunks
//      and the stream runs SEVEN steps ahead.  vmcnt retires in issue order, so a wave that waits for a fresh DMA
//      also waits for every older store; with a seven-step lead the waits of an item's first steps target
//      DMAs issued BEFORE the previous item's epilogue and count its loads and stores as younger
//      (vmcnt(63)): the stores get 2.5 us to drain instead of one step.
//      The nine taps of a chunk are walked DX-MAJOR: the ten 16-pixel row fragments (8 rows + 2 halo rows) of one dx
//      stay in registers for its three dy taps — 22 ds_read_b128 per 96 MFMAs (round 1: 48, tap-major: 36).  The
//      board runs this kernel at its power cap, so LDS bytes saved are clock gained (profiles/archive/r02_h_power.md).
//      The workgroup synchronises after steps 1, 3, 5, 7, 8 of a chunk only (barrier_after below).
//   3. RESIDUAL STREAM AS TWO 16-BIT PLANES (conv-B).  The fp32 residual value u is kept as
//      hi = (u + 0x8000) >> 16 (its bf16 rounding, ties away from zero) and lo = u & 0xffff: the pair restores u bit
//      for bit (u = ((hi - (lo >> 15)) << 16) | lo, all mod 2^16 / 2^32), `hi` IS the next convolution's bf16
//      operand, so the separate bf16 copy of round 1 is gone.
//   4. BLOCKED 16-BIT TENSORS.  The bf16 activations (t, hi) and the lo plane are stored [n][C/8][h][w][8]: an
//      8-channel block is a plane of 16-byte pixels.  The MFMA result gives a lane 8 channels of ONE pixel and its
//      neighbour lane the next pixel; channels-last (pixels 512 B apart) makes every 16-byte access its own cache
//      line, and the CU's texture addresser then spends a cycle per lane — ~80 cycles per 1-KiB store instruction,
//      more addresser time per conv-B item than the item has MFMA cycles.  Blocked, the 16 lanes of a pixel-row
//      segment read or write 256 contiguous bytes, and the input DMA's 64 lanes read runs of a halo row.
//
// Kept from round 1: weights as the MFMA's A operand with the row permutation that gives a lane 8 consecutive
// channels of one pixel (pack_conv_weights_bf16_host, perm16), the [channel group][pixel slot][16 B] input
// layout (conflict-free ds_read_b128), DMAs from inline asm with hand-counted vmcnt, persistent XCD-contiguous
// item walk, rotated item loop.  Bias is the accumulators' initial value.
#include <type_traits>

#include "conv3x3_bf16_common.h"

namespace dsen2 {

namespace {

using namespace bf16k;

constexpr int TH = 16, TW = 32;             // output pixels per item
constexpr int HH = TH + 2, HW = TW + 2;     // halo tile 18 x 34
constexpr int HALO = HH * HW;               // 612 pixel slots in use
constexpr int QS = 640;                     // slots per channel-group row: 10 DMA blocks of 64, = 0 mod 16
constexpr int NG = 4;                       // 16-byte channel groups per pixel and step (32 bf16 channels)
constexpr int IN_BYTES = NG * QS * 16;      // one input chunk buffer (40,960 B)
constexpr int IN_ROUNDS = 10;               // DMA rounds per issuing wave and chunk: wave q moves channel group q, 10 blocks of 64 slots
constexpr int WCH_BYTES = 32 * 128 * 2;     // one weight chunk: [4 k-groups][128 rows][16 B] = 8 KiB
constexpr int RING = 8;                     // weight ring slots
constexpr int LEAD = RING - 1;              // chunk c + LEAD is issued in step c
constexpr int THREADS = 512;                // 8 waves: (channel half) x (pixel quarter: 8 rows x 16 columns)
constexpr int MB = 4, PB = 8;               // per wave: 4 x 16 channels, 8 x 16 pixels (8 rows of one 16-column half)
constexpr int XR = PB + 2;                  // halo-row fragments kept per dx
constexpr size_t LDS_BYTES = (size_t)2 * IN_BYTES + (size_t)RING * WCH_BYTES + 256 * 4;
constexpr size_t LDS_BYTES_CHAIN = LDS_BYTES + 256 * 4;     // the chain kernel double-buffers the bias
// Cache policy of the once-per-block residual traffic (hi and lo loads, lo stores): nt (aux bit 1).  The stream is as
// large as the Infinity Cache and each value is touched once per block; same-box A/B on the VDSen2 bf16 bench:
// 16.03 k -> 16.39 k patches/s.  The hi stores keep the default policy: the next convolution reads them at once.
constexpr int kResPolicy = 2;
static_assert(QS >= HALO && QS % 16 == 0 && QS == 64 * IN_ROUNDS, "input chunk geometry");
static_assert(LDS_BYTES_CHAIN <= 160 * 1024, "LDS budget");

// ISSUING WAVES.  Waves w and w + 4 share a SIMD.  A DMA costs its wave address arithmetic plus ~100 cycles of issue,
// and a wave in that phase issues no MFMAs; with every wave issuing its share right after the barrier, both waves
// of a SIMD were in that phase together and the matrix pipe idled ~17 % of every step.  Waves 0-3 therefore issue
// ALL DMAs (two weight pieces and, in steps 0-4 of a chunk, two input rounds per step) while their SIMD partners 4-7 go straight
// to their MFMAs and never wait on vmcnt at all (publication = the issuing wave's wait + the step's barrier).
constexpr int W_PER_STEP = 2;               // weight pieces an issuing wave moves per step (its own and +4)
constexpr int rounds_in_step(int t) { return t < 5 ? 2 : 0; }
constexpr int first_round_of_step(int t) { return 2 * t; }
// BARRIERS.  (t below = index of a step inside its chunk, 0-8; the tap it computes is (dy, dx) = (t % 3, t / 3).)
// The workgroup synchronises after steps 1, 3, 5, 7 and 8 of every chunk — five barriers per nine steps, not
// nine (each one drains the matrix pipe of both waves of a SIMD).  What a barrier after step t must publish is every
// weight chunk read before the next barrier: fragments of chunk s+1 are read DURING step s, so the barrier after
// step s (index t in its chunk) covers chunks up to s + wait_depth(t) (3, or 2 after t = 7 because 8 has its own barrier).
// Ring reuse stays safe with LEAD = 7: the DMA issued at the start of step s overwrites chunk s-1, whose fragments
// were read during step s-2, and between any step s-2 and step s lies a barrier of that pattern.
constexpr bool barrier_after(int t) { return t == 8 || (t & 1) != 0; }
constexpr int wait_depth(int t) { return t == 7 ? 2 : 3; }
// vector-memory operations an issuing wave issues AFTER the weight DMAs of step s+depth-LEAD up to the end of step s
// (index t): the weight DMAs of the LEAD-depth steps up to s and those steps' input rounds (issued before the step's
// weight DMAs).  vmcnt(N) at the end of step s therefore retires the wave's pieces of weight chunk s+depth and
// everything older.
constexpr int younger_ops(int t, int depth, bool has_w, bool has_in) {
  int n = 0;
  for (int j = 0; j < LEAD - depth; ++j) n += (has_w ? W_PER_STEP : 0) + (has_in ? rounds_in_step((t - j + 9) % 9) : 0);
  return n;
}
// the input chunk staged in steps 0-4 is first read during step 8: by the end of step 7 everything up to step 4's last
// input round must have landed, i.e. all but the weight DMAs of steps 4-7
constexpr int younger_than_input(bool has_w) { return has_w ? 4 * W_PER_STEP : 0; }

}  // namespace

// EPI: kEpiRelu       out (bf16 NHWC) = relu(conv + bias)                                   conv-A
//      kEpiResidual   (aux, out2) = split(join(aux, out2) + res_scale * (conv + bias))      conv-B, in place on the planes
//      kEpiResidualF32  out (fp32 NHWC) = join(aux, out2) + res_scale * (conv + bias)       conv-B of the last block
// CINW = 32-bit words per input pixel (= F / 2).  ABL (diagnostic builds): timing-only ablation mask
// (1 no stores, 2 no residual loads, 4 no weight stream, 8 no input stream, 16 no barriers; 32 = full kernel + time stamps;
// 128 / 256 = only the issuing / only the compute-only waves skip their stores).  The closed experiments' masks (512 / 1536:
// epilogue traffic trickled under the next item; 2048: 8-bit lo plane) live in experiments/conv3x3_body16w_closed_masks.hip.txt.
//
// CHAIN (conv3x3_body16w_chain_kernel): ONE launch runs all 2d residual-block convolutions of a precision-1 network.
// A workgroup owns whole patches — every item of `patches_per_wg` consecutive images, layer after layer — so a layer's
// input was written by the same workgroup: no flag, no grid barrier, no cross-XCD visibility question; between two
// layers the workgroup only has to see its OWN stores (workgroup-scope release / acquire on one CU: a retired store
// is visible to every later load of the same CU).  The weight stream runs across the layer boundary (the packed
// weights of consecutive body layers are `layer_stride` bytes apart in one buffer), so the ring is always full.
// Two forms of the boundary:
//   * SEAMLESS (chain.seamless): the item loop just goes on — the last item of layer l stages chunk 0 of layer l+1's
//     first item like any next item's, the next bias arrives by LDS-DMA in the other half of a double buffer.  Valid
//     when no item reads what the item right before it in the workgroup's order wrote, except through input chunks
//     >= 4: an item's outputs are retired by every wave by the end of the NEXT item's chunk 2 (issuing waves: their
//     in-order counted waits, from chunk 0 on; compute-only waves: one vmcnt(0) there) and published by that step's
//     barrier — before chunk 3 stages chunk 4, and long before the last chunk stages the following item's chunk 0.  True
//     for >= 2 patches per workgroup (a patch's items of consecutive layers are then >= 2 positions apart) and for one
//     patch per workgroup at F = 256 (the only distance-1 pair is last item = (last tile, slab 1) -> first item, and
//     slab 1 is input chunks 4-7).  In-kernel stamps (tools/stamp_chain.py): 2.0-2.4 k cycles per boundary.
//   * DRAINED otherwise (F = 128 with one patch in THIS workgroup — a launch of one patch per workgroup, or the tail
//     workgroup of a batch that patches_per_wg does not divide; decided per workgroup from the patches it owns): every
//     wave retires everything (vmcnt(0)), barrier, the next layer's first input chunk is staged and awaited: 6-11 k
//     cycles per boundary.
// EPI is chosen per layer: conv-A (even) kEpiRelu hi -> t, conv-B (odd) kEpiResidual in place on (hi, lo), the last
// one kEpiResidualF32 -> out_f32 — three instantiations of the item loop in one kernel.  Same arithmetic per item as the
// per-layer kernels: same bits.  Diagnostic masks of the chain kernel: 1024 drained boundaries everywhere, 2048 (timing
// only) without the compute-only waves' wait, 4096 in-kernel stamps, 3 no epilogue traffic.

// X3 (precision 2, "bf16x3"): fp32-grade results on the bf16 matrix cores.  Every fp32 operand is the sum of two bf16
// numbers, x = xh + xl + O(2^-17 |x|) (xh = the value's bf16 rounding, xl = bf16(x - xh); weights split the same way at
// pack time), and a product is accumulated as xh*wh + xh*wl + xl*wh in fp32 (the xl*wl term is 2^-18 of the product):
// three MFMAs at 16 x the fp32 MFMA rate.  In this kernel that is nothing but a LONGER contraction: the input-chunk loop
// walks 3 * F/32 VIRTUAL chunks — chunk v = 3*cc + j reads real channels 32*cc .. 32*cc+31 of plane (xh, xh, xl)[j] against
// the weight planes (wh, wl, wh)[j], which pack_conv_weights_bf16x3_host lays out as a (3, 3, 3*F, F) kernel — so the step
// code, the DMA streams and their hand-counted waits are the bf16 kernel's, unchanged.  What differs:
//   * a 16-bit operand tensor has TWO planes per image, [n][2][F/8][h][w][8]: blocks 0 .. F/8-1 = hi, F/8 .. 2F/8-1 = lo
//     (conv-A reads the residual stream's (hi, xl), conv-B reads t's (hi, lo)); chunk v's block offset picks the plane;
//   * conv-A's epilogue writes relu(.) as (hi, lo) planes (RNE both); conv-B's reads the exact fp32 stream (hi, lo16: the
//     same two planes as precision 1, hi = ties-away bf16 rounding of the bit pattern) and writes hi, lo16 AND xl =
//     bf16(x - hi) — the stream stays exact fp32, xl exists only as conv-A's second operand plane;
//   * E_OPS: 32 stores (conv-A), 32 loads + 48 stores (conv-B; the first-chunk waits saturate at vmcnt(63): stricter).
// Accuracy (tests/test_gpu_bf16x3.py, against float64): whole DSen2_20 network rmse ~1e-5 in the normalised domain against
// 3e-7 (fp32) and 4e-3 (bf16 operands): inside the 1e-4 gate of BASELINE.md.
// X3 + CHAIN (conv3x3_body16w_x3_chain_kernel; capi.hip's forward_impl takes it for precision 2 wherever
// body16w_chain_patches_per_wg gives every CU whole patches): the boundary rules above were derived for one plane and 16 / 32
// epilogue operations; they carry over because every one of them is stated in CHUNKS of the item loop, and for X3 those are
// the virtual ones:
//   * retirement: an item's outputs are retired by the end of the next item's virtual chunk 2 and published by that step's
//     barrier, exactly as above (the stage-ahead distance is still one chunk of the loop = a third of a real chunk);
//   * what a distance-1 successor may read: at F = 256 the only distance-1 pair of a one-patch workgroup is (last tile, slab 1)
//     -> (tile 0, slab 0) of the next layer; slab 1 = real channels 128..255 = real chunks cc >= 4 = VIRTUAL chunks >= 12, in
//     both planes of a two-plane tensor (a plane is selected by j of v = 3*cc + j, never by cc) — later than the chunk-4
//     bound the rule needs.  F = 128 with one patch in the workgroup drains, as for precision 1 (ly_seamless is per workgroup);
//   * the counted waits: E_OPS is 32 (conv-A) / 80 (conv-B); a first-chunk wait that would count more than 63 operations
//     saturates at vmcnt(63) — it waits for MORE than it has to, never less — and E_FIRST takes the smaller of the two
//     alternating epilogues at a layer's first item, as for precision 1.
// tests/test_gpu_bf16x3.py::test_bf16x3_chain_kernel_equals_the_per_layer_kernels_bit_for_bit covers F = 128 and F = 256 with
// batches that patches_per_wg does not divide (tail workgroup with one patch, one or two tiles per layer).

template <int ABL, bool X3 = false>
__host__ __device__ constexpr int epilogue_ops(int epi) {
  // vector-memory operations of one epilogue (per wave): 16 groups of 8 channels x (stores + residual loads)
  // (diagnostic mask 128: the ISSUING waves 0-3 skip their epilogue stores, the compute-only waves keep theirs — the
  // waits exist in the issuers' code only, so the count is theirs; mask 256: the other way round)
  // X3: conv-A stores two planes (32), conv-B stores hi, lo16 and xl (48); kEpiResidualF32 is unchanged (32 fp32 stores)
  return ((ABL & (1 | 128)) ? 0 : (epi == kEpiRelu ? (X3 ? 32 : 16) : (X3 && epi == kEpiResidual ? 48 : 32))) +
         (epi != kEpiRelu && !(ABL & 2) ? 32 : 0);
}

template <int CINW, int COUT, int EPI0, int ABL, bool CHAIN, bool X3 = false>
__device__ __forceinline__ void body16w(const ConvParams p, const int n_items, const ChainArgs chain) {
  static_assert(!X3 || ABL == 0, "no diagnostic masks for bf16x3");
  constexpr int NCC = (X3 ? 3 : 1) * (CINW / 16);   // 32-channel chunks (X3: virtual chunks v = 3*cc + j, j = operand-plane pair)
  constexpr int IN_PLANES = X3 ? 2 : 1;             // 16-bit operand tensors: planes per image (hi | lo)
  constexpr unsigned LO_BLOCKS = CINW / 4;          // X3: block index of the lo plane inside an image (= F/8)
  constexpr int NCHUNK = NCC * 9;
  constexpr int NS = COUT / 128;
  constexpr bool kW = !(ABL & 4), kIn = !(ABL & 8);

  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* const in_s = reinterpret_cast<char*>(smem);                       // [2][4][QS][16 B]
  char* const w_s = in_s + 2 * IN_BYTES;                                  // [RING][4 k-groups][128 rows][16 B]
  float* const bias_s = reinterpret_cast<float*>(w_s + RING * WCH_BYTES);   // [COUT]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1;                 // 64-channel half of the slab
  const int wp = wave >> 1;                // pixel quarter of the tile: 8 rows x 16 columns
  const int wrow = 8 * (wp >> 1), wcol = 16 * (wp & 1);
  int l15 = lane & 15;
  int q4 = lane >> 4;

  // persistent schedule: logical ids remapped so that each XCD (blockIdx % 8) walks a contiguous run of items
  const int G = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  // Layer-invariant values the lambdas below use.  In a CHAIN they are passed through an empty asm statement at the
  // start of every layer (begin_layer): otherwise hipcc hoists what each of the six copies of the item loop derives
  // from them out of the layer loop and keeps all of it alive across all copies — more scalars than there are SGPRs.
  int W_ = p.w, H_ = p.h, TX_ = p.tiles_x, TPI_ = p.tiles_x * p.tiles_y;
  float RS_ = p.res_scale;
  // items this workgroup walks (per layer): item0, item0 + istep, ... (my_items of them)
  int item0, my_items;
  int my_imgs = 0;                           // CHAIN: whole patches this workgroup owns (the tail workgroup may own fewer)
  const int istep = CHAIN ? 1 : G;
  if constexpr (CHAIN) {
    const int first_img = lid * chain.patches_per_wg;
    const int imgs = p.n - first_img < chain.patches_per_wg ? p.n - first_img : chain.patches_per_wg;
    if (imgs <= 0) return;
    my_imgs = imgs;
    item0 = first_img * TPI_ * NS;
    my_items = imgs * TPI_ * NS;
  } else {
    if (lid >= n_items) return;
    item0 = lid;
    my_items = (n_items - lid + G - 1) / G;
  }
  size_t IMGPIX_ = (size_t)p.h * p.w;

  struct Tile { int img, ty0, tx0, slab; };
  auto tile_of = [&](int item) -> Tile {
    const int tile = item / NS;
    const int img = tile / TPI_;
    const int trem = tile - img * TPI_;
    const int tyi = trem / TX_;
    return Tile{img, tyi * TH, (trem - tyi * TX_) * TW, item - tile * NS};
  };

  // CHAIN: state of the current layer (set by the layer loop at the end of this function)
  float* bias_cur = bias_s;                 // LDS: the bias the accumulators start from ([2][COUT] in a chain)
  int ly_more = 0, ly_seamless = 0;         // another layer follows / its boundary is seamless
  unsigned bias_next_lds = 0, bias_next_so = 0;   // where the NEXT layer's bias goes (LDS byte address) / comes from (offset in w_rsrc)

  // The tensors of a convolution.  Per-layer kernel: the launch's parameters.  CHAIN: a function of the layer's
  // epilogue only (conv-A hi -> t; conv-B t -> (hi, lo) in place, or -> out_f32 for the last one), i.e. kernel arguments
  // inside each instantiation of the item loop — nothing per layer has to live in registers across it.
  auto in_of = [&](auto epi_c) -> const char* {
    if constexpr (CHAIN) return reinterpret_cast<const char*>(decltype(epi_c)::value == kEpiRelu ? chain.hi : chain.t);
    else return reinterpret_cast<const char*>(p.in);
  };
  auto out_of = [&](auto epi_c) -> char* {
    constexpr int e = decltype(epi_c)::value;
    if constexpr (CHAIN) return reinterpret_cast<char*>(e == kEpiRelu ? chain.t : e == kEpiResidual ? chain.hi : (void*)chain.out_f32);
    else return reinterpret_cast<char*>(p.out);
  };
  auto hi_of = [&]() -> char* { return reinterpret_cast<char*>(CHAIN ? chain.hi : (void*)const_cast<float*>(p.aux)); };
  auto lo_of = [&]() -> char* { return reinterpret_cast<char*>(CHAIN ? chain.lo : p.out2); };

  // ---- the two DMA streams ----
  const unsigned lds_in = lds_address(in_s), lds_w = lds_address(w_s);
  const int dq = wave & 3;                         // issuing wave q moves channel group q of every halo pixel
  __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpk), 0, 0, 0x00020000);   // set per staged item
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.wpk), 0,
      CHAIN ? (unsigned)chain.n_layers * chain.layer_stride : (unsigned)(NS * NCHUNK * WCH_BYTES), 0x00020000);
  unsigned in_plane_bytes = (unsigned)(IMGPIX_ * 16);   // one 8-channel block of one image
  int st_y0 = 0, st_x0 = 0;                        // origin of the tile being staged
  auto set_stage_item = [&](auto epi_c, int item) __attribute__((always_inline)) {
    const Tile t = tile_of(item);
    in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(in_of(epi_c)) + (size_t)t.img * IMGPIX_ * CINW * 4 * IN_PLANES, 0,
        (unsigned)(IMGPIX_ * CINW * 4 * IN_PLANES), 0x00020000);
    st_y0 = t.ty0;
    st_x0 = t.tx0;
  };
  // round r (0-9) of input chunk cc into buffer `buf`: 64 halo pixels of channel group dq
  auto issue_in = [&](int buf, int r, int cc) __attribute__((always_inline)) {
    // the per-lane offset is recomputed per round from an opaque copy of the lane id (a dozen VALU operations): ten
    // loop-invariant registers for the five rounds do not fit beside 128 accumulators
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int b = r;
    const int hp = 64 * b + ln;
    const int hy = (hp * 1928) >> 16, hx = hp - hy * HW;      // hp / 34 for hp < 640
    const int gy = st_y0 - 1 + hy, gx = st_x0 - 1 + hx;
    const bool inb = hp < HALO && (unsigned)gy < (unsigned)H_ && (unsigned)gx < (unsigned)W_;
    // blocked layout: 8-channel block (4*cc + dq) is a plane of 16-byte pixels, so the 64 lanes of a round read
    // runs of consecutive addresses (one run per halo row).  Out of range (any offset >= 2^31) reads zeros = the
    // convolution's padding; no branch
    const unsigned voff = ((unsigned)(__umul24(gy, W_) + gx) * 16u + in_plane_bytes * (unsigned)dq) | (inb ? 0u : 0x80000000u);
    const unsigned m0v = lds_in + buf * IN_BYTES + (dq * QS + 64 * b) * 16;
    unsigned blk = (unsigned)(4 * cc);
    if constexpr (X3) {
      // virtual chunk v = cc: real chunk v / 3, plane (hi, hi, lo)[v % 3]   (v < 96: v * 43691 >> 17 = v / 3)
      const unsigned real = ((unsigned)cc * 43691u) >> 17;
      blk = 4u * real + (((unsigned)cc - 3u * real) == 2u ? LO_BLOCKS : 0u);
    }
    const unsigned so = blk * in_plane_bytes;
    lds_dma(m0v, voff, in_rsrc, so);
  };
  // the next weight chunk of this workgroup's stream (8 wave instructions of 1 KiB, two per issuing wave) into the
  // next ring slot; the stream runs over item boundaries (CHAIN: and over layer boundaries) and, past the last item,
  // wraps to the first one (harmless)
  int wl_item = item0, wl_chunk = 0, wl_slot = 0;
  int wl_left = my_items;                          // CHAIN: items of the stream's layer still to come (this one included)
  unsigned wl_layer_off = 0;                       // CHAIN: byte offset of the stream's layer inside the weight buffer
  const unsigned w_voff = lane * 16;
  auto issue_w = [&]() __attribute__((always_inline)) {
    const unsigned so = (unsigned)(((wl_item % NS) * NCHUNK + wl_chunk) * WCH_BYTES + wave * 1024) + (CHAIN ? wl_layer_off : 0u);
    const unsigned m0v = lds_w + wl_slot * WCH_BYTES + wave * 1024;
    lds_dma(m0v, w_voff, w_rsrc, so);
    lds_dma(m0v + 4096u, w_voff, w_rsrc, so + 4096u);
    if constexpr (CHAIN) {
      // scalar selects only: a branch here would cut the nine-step body into basic blocks (see the step lambda)
      const bool item_done = wl_chunk == NCHUNK - 1;
      const bool layer_done = item_done && wl_left == 1;    // on to the next layer's first item (after the last layer: layer 0 again)
      const unsigned next_off = wl_layer_off + chain.layer_stride;
      wl_chunk = item_done ? 0 : wl_chunk + 1;
      wl_item = layer_done ? item0 : item_done ? wl_item + 1 : wl_item;
      wl_left = layer_done ? my_items : item_done ? wl_left - 1 : wl_left;
      wl_layer_off = !layer_done ? wl_layer_off : next_off >= (unsigned)chain.n_layers * chain.layer_stride ? 0u : next_off;
    } else {
      if (++wl_chunk == NCHUNK) {
        wl_chunk = 0;
        wl_item = wl_item + G < n_items ? wl_item + G : lid;
      }
    }
    wl_slot = wl_slot == RING - 1 ? 0 : wl_slot + 1;
  };

  // ---- per-lane operand addresses (bytes) ----
  // B operand (pixels): lane -> pixel column l15 of a 16-pixel row segment, channel group q4 of the chunk
  // A operand (weights): [k-group q4][row = wn*64 + 16*mb + l15][16 B]
  const int x_lane = (q4 * QS + wrow * HW + wcol + l15) * 16;
  const int w_lane = (q4 * 128 + wn * 64 + l15) * 16;

  // ---- prologue: first item's input chunk 0 (CHAIN: staged per layer, below), weight chunks 0 .. LEAD-1 ----
  // begin_layer: the layer's bias and its first item's input chunk 0 into LDS, everything in flight retired.  In a
  // CHAIN the same statement is what makes layer l's outputs (this workgroup's own stores) the input of layer l + 1:
  // every wave retires its stores (vmcnt(0)), the barrier orders them before the issuing waves' DMA reads.
  auto begin_layer = [&](auto epi_c, const float* bias) __attribute__((always_inline)) {
    if constexpr (CHAIN) {
      wait_vmcnt<0>();
      __syncthreads();
    }
    set_stage_item(epi_c, item0);
    if (wave < 4) {
      if constexpr (kIn) {
#pragma unroll
        for (int r = 0; r < IN_ROUNDS; ++r) issue_in(0, r, 0);
      }
    }
    if constexpr (CHAIN) {
      if (bias) {                     // the first layer's; every later one arrives by DMA during the layer before it
        int t = tid;                  // opaque copy: the LDS address is computed here, not kept (spilled) across the layers
        asm volatile("" : "+v"(t));
        if (t < COUT) bias_s[t] = bias[t];
      }
    } else {
      if (tid < COUT) bias_s[tid] = p.bias[tid];
    }
    wait_vmcnt<0>();
    __syncthreads();
  };
  if (wave < 4) {
    if constexpr (kW) {
#pragma unroll
      for (int c = 0; c < LEAD; ++c) issue_w();
    }
  }
  if constexpr (!CHAIN) begin_layer(std::integral_constant<int, EPI0>{}, p.bias);

  // Pixel fragments: XR = PB + 2 halo-row segments (rows wrow + 0..9 of the staged tile, 16 columns from wcol + dx)
  // serve the three taps dy = 0..2 of one dx: MFMA (mb, pb) of tap (dy, dx) multiplies x_row[pb + dy].
  f32x4 w_cur[MB], x_row[XR];
  int mf_slot = 0;
  auto read_rows = [&](const char* ib) {                    // dx = 0 of a chunk
#pragma unroll
    for (int r = 0; r < XR; ++r) x_row[r] = *reinterpret_cast<const f32x4*>(ib + x_lane + r * HW * 16);
  };
  auto read_w1 = [&](int mb, const char* wb) -> f32x4 {
    return *reinterpret_cast<const f32x4*>(wb + w_lane + mb * 256);
  };

  f32x4 acc[MB][PB];

  // diagnostic builds, ablation bit 32: s_memtime stamps of waves 0 and 7 of the first four workgroups, 32 per item
  int stamp_it = 0;
  auto stamp = [&](int k) __attribute__((always_inline)) {
    if constexpr ((ABL & 32) != 0) {
      if ((ABL & 64) != 0 && k != 0 && k != 19) return;      // 96: only the two loop-top stamps (clock measurement without the stamping overhead)
      if (p.diag && lid < 4 && (wave == 0 || wave == 7) && lane == 0)
        p.diag[(((size_t)lid * 2 + (wave == 7)) * 16 + (stamp_it & 15)) * 32 + k] =
            k == 19 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();      // slot 19: the 100 MHz counter
    }
  };

  // diagnostic mask 4096 (chain kernel): s_memtime of waves 0 and 7 of the first four workgroups at every item-loop top
  // (k = 0) and after every epilogue (k = 1): slot [workgroup][wave 0 / 7][layer][iteration 0..7][k]
  int stamp_layer = 0;
  auto stampc = [&](int k) __attribute__((always_inline)) {
    if constexpr (CHAIN && (ABL & 4096) != 0) {
      if (p.diag && lid < 4 && (wave == 0 || wave == 7) && lane == 0 && stamp_it < 8 && stamp_layer < 64)
        p.diag[((((size_t)lid * 2 + (wave == 7)) * 64 + stamp_layer) * 8 + stamp_it) * 2 + k] = __builtin_amdgcn_s_memtime();
    }
  };

  // ---- epilogue of one item: lane = pixel (row wrow + pb, column wcol + l15), group (pr, pb) =
  // 8 consecutive channels slab*128 + wn*64 + 32*pr + 8*q4 held by accumulators 2*pr and 2*pr+1 ----
  auto epilogue = [&](auto epi_c, int item, bool valid) __attribute__((always_inline)) {
    constexpr int EPI = decltype(epi_c)::value;
    const Tile t = tile_of(item);
    const int ch8 = t.slab * 128 + wn * 64 + 8 * q4;
    const int ex = t.tx0 + wcol + l15, ey = t.ty0 + wrow;
    // Byte offset of group (pr, pb) inside its image, branch-free.  16-bit tensors are BLOCKED: 8-channel block k of
    // an image is a plane [h][w] of 16-byte pixels, so the 16 lanes of a pixel-row segment touch 256 contiguous
    // bytes and the texture addresser coalesces them four lanes at a time (with channels-last pixels 512 B apart it
    // spent one cycle per lane: 80 cycles per store instruction).  A pixel outside the image (ragged tile; the dummy
    // epilogue before the first item) gets bit 31 set = out of the descriptor's range: every load and store is
    // always ISSUED, which is what the hand-counted waits of the next item's first steps rely on.
    const unsigned blk0 = (unsigned)(ch8 >> 3);                       // + 4*pr
    const unsigned base_pix = (unsigned)(ey * W_ + ex);
    const unsigned bad_all = valid ? 0u : 0x80000000u;
    auto bad_of = [&](int pb) -> unsigned {
      const int row = ey + pb, col = ex;
      return bad_all | (row < H_ ? 0u : 0x80000000u) | (col < W_ ? 0u : 0x80000000u);
    };
    auto plane_off = [&](int pr, int pb) -> unsigned {                // 16-bit blocked tensors
      return (((blk0 + 4u * pr) * (unsigned)IMGPIX_ + base_pix + (unsigned)(pb * W_)) * 16u & 0x7fffffffu) | bad_of(pb);
    };
    // X3: the same group in the SECOND plane of a two-plane tensor (blocks COUT/8 .. 2*COUT/8-1 of the image)
    auto plane2_off = [&](int pr, int pb) -> unsigned {
      return (((blk0 + 4u * pr + (unsigned)(COUT / 8)) * (unsigned)IMGPIX_ + base_pix + (unsigned)(pb * W_)) * 16u & 0x7fffffffu) | bad_of(pb);
    };
    // X3: bf16(x - hi) for the two fp32 values whose bf16 roundings are packed in `h` (low half = first value): the second
    // operand plane.  x - hi is exact in fp32 (hi is within one bf16 ulp of x), its RNE rounding leaves 2^-17 |x|.
    auto lo_of_pair = [&](float x0, float x1, unsigned h) -> unsigned {
      return pack_bf16(x0 - __builtin_bit_cast(float, h << 16), x1 - __builtin_bit_cast(float, h & 0xffff0000u));
    };
    auto nhwc_f32_off = [&](int pr, int pb) -> unsigned {             // fp32 channels-last tensor (kEpiResidualF32)
      return (((base_pix + (unsigned)(pb * W_)) * (unsigned)COUT + (unsigned)(ch8 + 32 * pr)) * 4u & 0x7fffffffu) | bad_of(pb);
    };
    const size_t img_elems = IMGPIX_ * COUT;
    if constexpr (EPI == kEpiRelu) {
      const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          out_of(epi_c) + (size_t)t.img * img_elems * 2 * IN_PLANES, 0, (unsigned)(img_elems * 2 * IN_PLANES), 0x00020000);
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          f32x4 v0 = acc[2 * pr][pb], v1 = acc[2 * pr + 1][pb];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v0[e] = fmaxf(v0[e], 0.f);
            v1[e] = fmaxf(v1[e], 0.f);
          }
          const u32x4 hv = {pack_bf16(v0[0], v0[1]), pack_bf16(v0[2], v0[3]), pack_bf16(v1[0], v1[1]), pack_bf16(v1[2], v1[3])};
          if constexpr (X3) {
            // relu(conv + b) as two bf16 planes: hi = its RNE rounding, lo = bf16(value - hi)
            const u32x4 lv = {lo_of_pair(v0[0], v0[1], hv[0]), lo_of_pair(v0[2], v0[3], hv[1]), lo_of_pair(v1[0], v1[1], hv[2]),
                              lo_of_pair(v1[2], v1[3], hv[3])};
            __builtin_amdgcn_raw_buffer_store_b128(hv, out_rsrc, plane_off(pr, pb), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(lv, out_rsrc, plane2_off(pr, pb), 0, 0);
          } else if constexpr ((ABL & 384) != 0) {
            if ((wave < 4) == ((ABL & 128) != 0)) asm volatile("" ::"v"(hv)); else __builtin_amdgcn_raw_buffer_store_b128(hv, out_rsrc, plane_off(pr, pb), 0, 0);
          } else if constexpr (!(ABL & 1))
            __builtin_amdgcn_raw_buffer_store_b128(hv, out_rsrc, plane_off(pr, pb), 0, 0);
          else
            asm volatile("" ::"v"(hv));
          if (pr == 1) __builtin_amdgcn_sched_barrier(0);     // bounds the packed values in flight (registers)
        }
    } else {
      const auto hi_rsrc = __builtin_amdgcn_make_buffer_rsrc(          // X3: the stream's operand tensor (hi | xl planes)
          hi_of() + (size_t)t.img * img_elems * 2 * IN_PLANES, 0, (unsigned)(img_elems * 2 * IN_PLANES), 0x00020000);
      const auto lo_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          lo_of() + (size_t)t.img * img_elems * 2, 0, (unsigned)(img_elems * 2), 0x00020000);
      const auto f32_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          out_of(epi_c) + (EPI == kEpiResidualF32 ? (size_t)t.img * img_elems * 4 : 0), 0,
          EPI == kEpiResidualF32 ? (unsigned)(img_elems * 4) : 0, 0x00020000);
      // pass j = rows wrow + 2j, 2j+1: 4 groups (2 rows x 2 channel pairs).  Residual loads run two passes
      // ahead of the stores in issue order (L0 L1 | C0 L2 S0 | C1 L3 S1 | C2 S2 | C3 S3): only the last pass's
      // loads are younger than a store (pass 0's), so vmcnt's in-order retirement exposes one store drain, not four
      u32x4 rh[4][4], rl[4][4];
      auto load_pass = [&](int j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int pb = 2 * j + (g >> 1), pr = g & 1;
          if constexpr (!(ABL & 2)) {
            const unsigned eo = plane_off(pr, pb);
            rh[j][g] = __builtin_amdgcn_raw_buffer_load_b128(hi_rsrc, eo, 0, kResPolicy);
            rl[j][g] = __builtin_amdgcn_raw_buffer_load_b128(lo_rsrc, eo, 0, kResPolicy);
          } else {
            rh[j][g] = rl[j][g] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
          }
        }
      };
      auto finish_pass = [&](int j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int pb = 2 * j + (g >> 1), pr = g & 1;
          const f32x4 c0 = acc[2 * pr][pb], c1 = acc[2 * pr + 1][pb];
          const float cv[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
          unsigned ov[8];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            unsigned u0, u1;
            join2(rh[j][g][k], rl[j][g][k], u0, u1);
            // x + 0.1 * (conv + b): two roundings like keras (-ffp-contract=off)
            ov[2 * k] = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, u0) + cv[2 * k] * RS_);
            ov[2 * k + 1] = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, u1) + cv[2 * k + 1] * RS_);
          }
          if constexpr (EPI == kEpiResidual) {
            const unsigned eo = plane_off(pr, pb);
            u32x4 oh, ol;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              unsigned h_, l_;
              split2(ov[2 * k], ov[2 * k + 1], h_, l_);
              oh[k] = h_;
              ol[k] = l_;
            }
            if constexpr (X3) {
              // the exact fp32 stream (hi, lo16) as in precision 1, plus xl = bf16(x - hi): conv-A's second operand plane
              u32x4 ox;
#pragma unroll
              for (int k = 0; k < 4; ++k)
                ox[k] = lo_of_pair(__builtin_bit_cast(float, ov[2 * k]), __builtin_bit_cast(float, ov[2 * k + 1]), oh[k]);
              __builtin_amdgcn_raw_buffer_store_b128(oh, hi_rsrc, eo, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b128(ox, hi_rsrc, plane2_off(pr, pb), 0, 0);
              __builtin_amdgcn_raw_buffer_store_b128(ol, lo_rsrc, eo, 0, kResPolicy);
            } else if constexpr ((ABL & 384) != 0) {
              if ((wave < 4) == ((ABL & 128) != 0)) {
                asm volatile("" ::"v"(oh), "v"(ol));
              } else {
                __builtin_amdgcn_raw_buffer_store_b128(oh, hi_rsrc, eo, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(ol, lo_rsrc, eo, 0, kResPolicy);
              }
            } else if constexpr (!(ABL & 1)) {
              __builtin_amdgcn_raw_buffer_store_b128(oh, hi_rsrc, eo, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b128(ol, lo_rsrc, eo, 0, kResPolicy);
            } else {
              asm volatile("" ::"v"(oh), "v"(ol));
            }
          } else {
            const u32x4 o0 = {ov[0], ov[1], ov[2], ov[3]}, o1 = {ov[4], ov[5], ov[6], ov[7]};
            const unsigned eo = nhwc_f32_off(pr, pb);
            if constexpr (!(ABL & 1)) {
              // immediate soffset only (store-data hazard of buffer_store_dwordx4 with an SGPR soffset, experiments/README.md)
              __builtin_amdgcn_raw_buffer_store_b128(o0, f32_rsrc, eo, 0, 0);
              __builtin_amdgcn_raw_buffer_store_b128(o1, f32_rsrc, eo + 16u, 0, 0);
            } else {
              asm volatile("" ::"v"(o0), "v"(o1));
            }
          }
        }
      };
      if constexpr (X3) {
        // one pass of look-ahead (L0 | L1 C0 S0 | L2 C1 S1 | L3 C2 S2 | C3 S3): the second operand plane costs registers, and an
        // item's 3 x longer contraction makes the epilogue a third as important
        load_pass(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          __builtin_amdgcn_sched_barrier(0);
          if (j + 1 < 4) load_pass(j + 1);
          __builtin_amdgcn_sched_barrier(0);
          finish_pass(j);
        }
      } else {
      load_pass(0);
      load_pass(1);
      stamp(14);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        __builtin_amdgcn_sched_barrier(0);
        if (j + 2 < 4) load_pass(j + 2);
        __builtin_amdgcn_sched_barrier(0);
        finish_pass(j);
        stamp(15 + j);
      }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // Two copies of the item loop: ISSUER (waves 0-3: DMAs + hand-counted waits) and worker (waves 4-7).
  // Rotated item loop: iteration `it` first writes out item it-1 (the first iteration issues the same loads and
  // stores with out-of-range offsets, so every path into an item's first steps has issued E_OPS operations),
  // then runs item `it`'s NCHUNK steps; one extra iteration writes the last item.
  auto run = [&](auto issuer_c, auto epi_c) __attribute__((always_inline)) {
  constexpr bool ISSUER = decltype(issuer_c)::value;
  constexpr int E_OPS = epilogue_ops<ABL, X3>(decltype(epi_c)::value);
  // CHAIN: the layer before / after this one (conv-A and conv-B alternate); what is still in flight when a layer's
  // FIRST item starts is the other kind's epilogue — its waits may count no more than the smaller of the two
  constexpr int EPI_OTHER = decltype(epi_c)::value == kEpiRelu ? kEpiResidual : kEpiRelu;
  constexpr int E_FIRST = !CHAIN || epilogue_ops<ABL, X3>(EPI_OTHER) > E_OPS ? E_OPS : epilogue_ops<ABL, X3>(EPI_OTHER);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) acc[mb][pb] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int it = 0; it <= my_items; ++it) {
    stamp_it = it;
    stamp(0);
    stamp(19);
    // (CHAIN: no dummy epilogue before a layer's first item — begin_layer has retired everything older than the
    // item's own DMAs, so the first-chunk waits, which count E_OPS operations that are then not there, have nothing
    // older left to protect)
    stampc(0);
    if (!CHAIN || it > 0) epilogue(epi_c, it > 0 ? item0 + (it - 1) * istep : item0, it > 0);
    __builtin_amdgcn_sched_barrier(0);
    stamp(1);
    stampc(1);
    if (it == my_items) break;
    const int item = item0 + it * istep;
    const bool have_next_item = it + 1 < my_items;
    {
      // accumulators start at the bias of their channels: acc[2*pr + e][.][r] <-> channel ch8 + 32*pr + 4*e + r
      const int ch8 = (item % NS) * 128 + wn * 64 + 8 * q4;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias_cur + ch8 + 32 * (mb >> 1) + 4 * (mb & 1));
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) acc[mb][pb] = b;
      }
    }
    // fragments of the item's first step (not prefetched across the epilogue: it needs the registers)
    {
      const char* const wb = w_s + mf_slot * WCH_BYTES;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) w_cur[mb] = read_w1(mb, wb);
      read_rows(in_s);
    }
    // These reads of chunk 0 happen IN the item's first step, not one step earlier like every other fragment read,
    // and step 1's weight DMA reuses chunk 0's ring slot with no barrier after step 0: synchronise here, once per item.
    if constexpr (!(ABL & 16)) __syncthreads();
    stamp(2);
    if constexpr (CHAIN) {
      // the next layer's bias into the other half of the double buffer (read two hundred steps from now)
      if (ISSUER && it == 0 && ly_more != 0 && wave < COUT / 64) lds_dma_dword(bias_next_lds, lane * 4, w_rsrc, bias_next_so);
    }
    const int first_item = __builtin_amdgcn_readfirstlane(it == 0 ? 1 : 0);   // an SGPR for the asm statements below

    // ONE copy of the nine-step body for every input chunk (a separate copy for the item's first chunk makes the
    // register allocator permute all 32 accumulators between the two copies and spill); the first chunk's five
    // epilogue-aware waits are a scalar branch
    auto do_cc = [&](const int cc) __attribute__((always_inline)) {
      const char* const ib = in_s + (cc & 1) * IN_BYTES;
      const char* const ib_next = in_s + ((cc + 1) & 1) * IN_BYTES;
      // staged into ib_next during this cc: (this item, cc+1), or on the last cc the NEXT item's chunk 0 (on the
      // very last item: its own chunk 0 again, which nobody reads — the operation count stays the same)
      const bool last_cc = cc == NCC - 1;
      const int in_cc = last_cc ? 0 : cc + 1;
      if (ISSUER && last_cc) {
        if (have_next_item) set_stage_item(epi_c, item + istep);
        else if constexpr (CHAIN) {       // seamless boundary: chunk 0 of the next layer's first item
          if (ly_seamless != 0 && ly_more != 0) set_stage_item(std::integral_constant<int, EPI_OTHER>{}, item0);
        }
      }
      auto step = [&](auto st_c) __attribute__((always_inline)) {
        constexpr int st = decltype(st_c)::value;        // index of the step inside its chunk
        const int nx_slot = mf_slot == RING - 1 ? 0 : mf_slot + 1;
        const char* const wb_nx = w_s + nx_slot * WCH_BYTES;
        // this step's DMAs: input rounds first, then the weight chunk LEAD steps ahead
        if constexpr (ISSUER && kIn) {
#pragma unroll
          for (int r = 0; r < rounds_in_step(st); ++r) issue_in((cc + 1) & 1, first_round_of_step(st) + r, in_cc);
        }
        if constexpr (ISSUER && kW) issue_w();
        // Step st of a chunk computes tap (dy, dx) = (st % 3, st / 3): dx-major, so that the XR row fragments of one dx
        // serve three steps (22 fragment reads per three steps instead of 36).  Fragments are refilled in place as
        // soon as their last MFMA of this dx has issued: row 0 after (MB-1, 0) of dy = 0, row 1 after (MB-1, 0) of
        // dy = 1, row pb + 2 after (MB-1, pb) of dy = 2; weight fragment mb after its 8 MFMAs.  The last dx of a chunk
        // refills from the NEXT chunk's buffer, which is only complete after step 7's barrier: its rows 0 and 1 are
        // read at the start of step 8.  (An item's last step reads the next item's first fragments too; they are
        // read again after the epilogue, which needs the registers.)
        constexpr int dy = st % 3, dx = st / 3;
        const char* const xb_nx = (dx < 2 ? ib + (dx + 1) * 16 : ib_next) + x_lane;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (st == 8) {
          x_row[0] = *reinterpret_cast<const f32x4*>(xb_nx);
          x_row[1] = *reinterpret_cast<const f32x4*>(xb_nx + HW * 16);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
          for (int pb = 0; pb < PB; ++pb) {
            acc[mb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w_cur[mb]),
                                                                  __builtin_bit_cast(bf16x8, x_row[pb + dy]),
                                                                  acc[mb][pb], 0, 0, 0);
            if (mb == MB - 1 && (dy == 2 || (pb == 0 && dx < 2))) {
              const int r = dy == 2 ? pb + 2 : dy;
              __builtin_amdgcn_sched_barrier(0);
              x_row[r] = *reinterpret_cast<const f32x4*>(xb_nx + r * HW * 16);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          w_cur[mb] = read_w1(mb, wb_nx);
          __builtin_amdgcn_sched_barrier(0);
        }
        mf_slot = nx_slot;
        __builtin_amdgcn_sched_barrier(0);
        // ISSUER, before a barrier: retire this wave's pieces of the weight chunk wait_depth steps ahead and everything
        // older; while that chunk was issued before the previous epilogue (the item's first steps), the epilogue's
        // loads and stores are younger than it and stay in flight.  Workers have no DMA of their own to wait for.
        if constexpr (barrier_after(st)) {
          if constexpr (ISSUER) {
            constexpr int kD = wait_depth(st);
            constexpr int kN0 = younger_ops(st, kD, kW, kIn);
            constexpr int kN = (st == 7 && kIn && younger_than_input(kW) < kN0) ? younger_than_input(kW) : kN0;
            constexpr int kNE = kN + E_OPS < 63 ? kN + E_OPS : 63;
            constexpr int kNF = kN + E_FIRST < 63 ? kN + E_FIRST : 63;
            if constexpr (st + kD < LEAD && kNE != kN && kNF != kNE) {
              // CHAIN, a layer's first item: the epilogue in flight is the previous layer's (fewer operations)
              asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc0 .Ldsen2_n%=\n\ts_cmp_eq_u32 %1, 0\n\ts_cbranch_scc1 .Ldsen2_w%=\n\t"
                           "s_waitcnt vmcnt(%4)\n\ts_branch .Ldsen2_e%=\n"
                           ".Ldsen2_w%=:\n\ts_waitcnt vmcnt(%3)\n\ts_branch .Ldsen2_e%=\n"
                           ".Ldsen2_n%=:\n\ts_waitcnt vmcnt(%2)\n.Ldsen2_e%=:"
                           ::"s"(cc), "s"(first_item), "n"(kN), "n"(kNE), "n"(kNF) : "memory", "scc");
            } else if constexpr (st + kD < LEAD && kNE != kN) {
              // vmcnt(kNE) in the item's first chunk, vmcnt(kN) otherwise.  The scalar branch lives inside ONE asm
              // statement so that the nine-step body stays a single basic block (split into blocks, hipcc's register
              // allocator shuffles the accumulators between them and spills into the DMA-counted vmcnt stream).
              asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Ldsen2_w%=\n\ts_waitcnt vmcnt(%1)\n\ts_branch .Ldsen2_e%=\n"
                           ".Ldsen2_w%=:\n\ts_waitcnt vmcnt(%2)\n.Ldsen2_e%=:"
                           ::"s"(cc), "n"(kN), "n"(kNE) : "memory", "scc");
            } else {
              wait_vmcnt<kN>();
            }
          }
          if constexpr (CHAIN && !ISSUER && st == 8 && !(ABL & 2048)) {
            // compute-only waves never wait on vmcnt; in a chain their stores are the next layer's input, so once per
            // item — at the end of its chunk 2 — they retire the previous item's epilogue (issued 27 steps ago).
            // (diagnostic mask 2048, timing only: without this wait)
            asm volatile("s_cmp_eq_u32 %0, 2\n\ts_cbranch_scc0 .Ldsen2_k%=\n\ts_waitcnt vmcnt(0)\n.Ldsen2_k%=:" ::"s"(cc) : "memory", "scc");
          }
          if constexpr (!(ABL & 16)) __syncthreads();
        }
        if constexpr ((ABL & 32) != 0) {
          if (cc == 0) stamp(3 + st);
          if (st == 8 && cc == 1) stamp(12);
          if (st == 8 && cc == NCC - 1) stamp(13);
        }
      };
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
      step(std::integral_constant<int, 4>{});
      step(std::integral_constant<int, 5>{});
      step(std::integral_constant<int, 6>{});
      step(std::integral_constant<int, 7>{});
      step(std::integral_constant<int, 8>{});
    };
#pragma unroll 1
    for (int cc = 0; cc < NCC; ++cc) do_cc(cc);
  }
  };   // run
  if constexpr (!CHAIN) {
    if (wave < 4)
      run(std::true_type{}, std::integral_constant<int, EPI0>{});
    else
      run(std::false_type{}, std::integral_constant<int, EPI0>{});
  } else {
    const unsigned lds_bias = lds_address(bias_s);
    const unsigned bias_delta = (unsigned)(reinterpret_cast<const char*>(p.bias) - reinterpret_cast<const char*>(p.wpk));
    // Seamless boundaries are a property of what THIS workgroup owns, not of the launch: the tail workgroup of a batch
    // that patches_per_wg does not divide may hold a single patch, and at F = 128 that is the case that must drain (its
    // layer l+1 first item would stage what its own layer-l last item has not stored yet).  Uniform per workgroup.
    // (diagnostic mask 1024: drained boundaries even where seamless ones are valid, A/B)
    ly_seamless = (ABL & 1024) ? 0 : (chain.seamless != 0 && (my_imgs >= 2 || COUT == 256)) ? 1 : 0;
#pragma unroll 1
    for (int l = 0; l < chain.n_layers; ++l) {
      // Layer-invariant values pass through an empty asm statement at the start of every layer (see their definition).
      // (two statements: hipcc treats EVERY output of an asm statement as divergent when one of them is a VGPR)
      asm volatile("" : "+s"(W_), "+s"(H_), "+s"(TX_), "+s"(TPI_), "+s"(RS_), "+s"(IMGPIX_), "+s"(in_plane_bytes));
      asm volatile("" : "+v"(l15), "+v"(q4));
      stamp_layer = l;
      ly_more = l + 1 < chain.n_layers ? 1 : 0;
      bias_cur = bias_s + (l & 1) * COUT;
      bias_next_lds = lds_bias + (unsigned)(((l + 1) & 1) * COUT * 4 + wave * 256);
      bias_next_so = bias_delta + (unsigned)(l + 1) * chain.layer_stride + (unsigned)(wave * 256);
      auto layer = [&](auto epi_c) __attribute__((always_inline)) {
        if (l == 0) begin_layer(epi_c, p.bias);
        else if (ly_seamless == 0) begin_layer(epi_c, nullptr);
        if (wave < 4) run(std::true_type{}, epi_c);
        else run(std::false_type{}, epi_c);
      };
      if ((l & 1) == 0) layer(std::integral_constant<int, kEpiRelu>{});
      else if (l + 1 < chain.n_layers) layer(std::integral_constant<int, kEpiResidual>{});
      else layer(std::integral_constant<int, kEpiResidualF32>{});
    }
  }
  wait_vmcnt<0>();       // no DMA may still be writing this workgroup's LDS when it is released
}

template <int CINW, int COUT, int EPI, int ABL>
__global__ __launch_bounds__(THREADS, 2) void conv3x3_body16w_kernel(const ConvParams p, const int n_items) {
  body16w<CINW, COUT, EPI, ABL, false>(p, n_items, ChainArgs{});
}

// precision 2 (bf16x3): p.in = a two-plane 16-bit tensor, weights packed by pack_conv_weights_bf16x3_host
template <int CINW, int COUT, int EPI>
__global__ __launch_bounds__(THREADS, 2) void conv3x3_body16w_x3_kernel(const ConvParams p, const int n_items) {
  body16w<CINW, COUT, EPI, 0, false, true>(p, n_items, ChainArgs{});
}

// p.wpk / p.bias: the FIRST body layer's packed weights / bias; p.in, p.out, p.aux, p.out2 are set per layer
template <int CINW, int COUT, int ABL>
__global__ __launch_bounds__(THREADS, 2) void conv3x3_body16w_chain_kernel(const ConvParams p, const ChainArgs chain) {
  body16w<CINW, COUT, kEpiRelu, ABL, true>(p, 0, chain);
}
// ... of a precision-2 network: chain.hi = the stream's operand tensor (hi | xl planes), chain.t two planes as well
template <int CINW, int COUT>
__global__ __launch_bounds__(THREADS, 2) void conv3x3_body16w_x3_chain_kernel(const ConvParams p, const ChainArgs chain) {
  body16w<CINW, COUT, kEpiRelu, 0, true, true>(p, 0, chain);
}

template <int CINW, int COUT, int EPI, int ABL = 0>
static hipError_t launch_body16w_one(ConvParams p, hipStream_t stream, int grid_cap) {
  auto kern = conv3x3_body16w_kernel<CINW, COUT, EPI, ABL>;
  static KernelOnce once;
  int cus = 0;
  hipError_t e = once.prepare(reinterpret_cast<const void*>(kern), LDS_BYTES, &cus);
  if (e != hipSuccess) return e;
  // per-image buffer descriptors: byte offsets below 2^31 (bit 31 marks a pixel outside the image)
  if ((size_t)p.h * p.w * COUT >= ((size_t)1 << 29)) return hipErrorInvalidValue;
  p.tiles_x = (p.w + TW - 1) / TW;
  p.tiles_y = (p.h + TH - 1) / TH;
  const long long items = (long long)p.n * p.tiles_x * p.tiles_y * (COUT / 128);
  if (items <= 0 || items > 0x7fffffffLL) return hipErrorInvalidValue;
  int grid = (int)(items < cus ? items : cus);
  if (grid_cap > 0 && grid_cap < grid) grid = grid_cap;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), LDS_BYTES, stream, p, (int)items);
  return hipGetLastError();
}

template <int CINW, int COUT, int EPI>
static hipError_t launch_body16w_x3_one(ConvParams p, hipStream_t stream) {
  auto kern = conv3x3_body16w_x3_kernel<CINW, COUT, EPI>;
  static KernelOnce once;
  int cus = 0;
  hipError_t e = once.prepare(reinterpret_cast<const void*>(kern), LDS_BYTES, &cus);
  if (e != hipSuccess) return e;
  // per-image descriptors of the two-plane tensors (4 bytes per value): byte offsets below 2^31
  if ((size_t)p.h * p.w * COUT >= ((size_t)1 << 29)) return hipErrorInvalidValue;
  p.tiles_x = (p.w + TW - 1) / TW;
  p.tiles_y = (p.h + TH - 1) / TH;
  const long long items = (long long)p.n * p.tiles_x * p.tiles_y * (COUT / 128);
  if (items <= 0 || items > 0x7fffffffLL) return hipErrorInvalidValue;
  const int grid = (int)(items < cus ? items : cus);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), LDS_BYTES, stream, p, (int)items);
  return hipGetLastError();
}

template <int F>
static hipError_t launch_body16w_x3_feat(const ConvParams& p, int epilogue, hipStream_t stream) {
  if (epilogue == kEpiRelu) return launch_body16w_x3_one<F / 2, F, kEpiRelu>(p, stream);
  if (epilogue == kEpiResidual) return launch_body16w_x3_one<F / 2, F, kEpiResidual>(p, stream);
  if (epilogue == kEpiResidualF32) return launch_body16w_x3_one<F / 2, F, kEpiResidualF32>(p, stream);
  return hipErrorInvalidValue;
}

hipError_t launch_conv3x3_body16w_x3(const ConvParams& p, int feat, int epilogue, hipStream_t stream) {
  if (!p.in || !p.wpk || !p.bias) return hipErrorInvalidValue;
  if (epilogue != kEpiRelu && (!p.aux || !p.out2)) return hipErrorInvalidValue;
  if (epilogue != kEpiResidual && !p.out) return hipErrorInvalidValue;
  if (feat == 128) return launch_body16w_x3_feat<128>(p, epilogue, stream);
  if (feat == 256) return launch_body16w_x3_feat<256>(p, epilogue, stream);
  return hipErrorInvalidValue;
}

template <int CINW, int COUT, int ABL = 0, bool X3 = false>
static hipError_t launch_body16w_chain_one(ConvParams p, ChainArgs c, hipStream_t stream) {
  auto kern = [] {
    if constexpr (X3) return conv3x3_body16w_x3_chain_kernel<CINW, COUT>;
    else return conv3x3_body16w_chain_kernel<CINW, COUT, ABL>;
  }();
  static KernelOnce once;
  int cus = 0;
  hipError_t e = once.prepare(reinterpret_cast<const void*>(kern), LDS_BYTES_CHAIN, &cus);
  if (e != hipSuccess) return e;
  if ((size_t)p.h * p.w * COUT >= ((size_t)1 << 29)) return hipErrorInvalidValue;
  p.tiles_x = (p.w + TW - 1) / TW;
  p.tiles_y = (p.h + TH - 1) / TH;
  // a layer = its packed weights, then (within layer_stride) its bias: the kernel addresses both through one descriptor
  const long long bias_delta = reinterpret_cast<const char*>(p.bias) - reinterpret_cast<const char*>(p.wpk);
  if (bias_delta < (long long)(COUT / 128) * (X3 ? 3 : 1) * (CINW / 16) * 9 * WCH_BYTES || bias_delta + COUT * 4 > (long long)c.layer_stride)
    return hipErrorInvalidValue;
  c.patches_per_wg = body16w_chain_patches_per_wg(p.n, p.h, p.w, COUT, cus);
  if (c.n_layers <= 0) return hipErrorInvalidValue;
  if (c.patches_per_wg <= 0) return hipErrorNotSupported;       // the per-layer kernels keep more CUs busy for this batch
  // the weight descriptor spans all layers: 32-bit byte offsets
  if ((unsigned long long)c.n_layers * c.layer_stride >= 0xffffffffull) return hipErrorInvalidValue;
  const int grid = (p.n + c.patches_per_wg - 1) / c.patches_per_wg;
  c.seamless = (c.patches_per_wg >= 2 || COUT == 256) ? 1 : 0;     // see the kernel's header: when no drain is needed
  hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), LDS_BYTES_CHAIN, stream, p, c);
  return hipGetLastError();
}

// Whole patches per workgroup for the chain kernel, or 0 when a chain would leave CUs idle that the per-layer kernels
// use: the chain hands out images, the per-layer launch items; chain only if it needs no more item rounds per layer.
// ... and only for patches of at most 8 tiles (64 x 64): measured (tools/chain_vs_layerwise_probe.py,
// profiles/r04_k_chain_vs_layerwise.txt) the chain is 1-3 % ahead of the per-layer launches up to there and 1-5 % BEHIND them
// at 96^2 (F = 256), 128^2 and 192^2 patches in every mode — with 32+ items per patch and layer the launch boundaries it
// removes are a per-mille of the layer's time, and a workgroup walking its own patches alone loses the L2 sharing of halo
// rows that neighbouring workgroups get when a layer's items are handed out in order.
constexpr int kChainMaxTilesPerPatch = 8;

int body16w_chain_patches_per_wg(int n, int h, int w, int feat, int cus) {
  if (n <= 0 || cus <= 0) return 0;
  if ((long long)((w + TW - 1) / TW) * ((h + TH - 1) / TH) > kChainMaxTilesPerPatch) return 0;
  const long long ipp = (long long)((w + TW - 1) / TW) * ((h + TH - 1) / TH) * (feat / 128);    // items per patch
  const int ppw = (n + cus - 1) / cus;
  const long long rounds_layerwise = (n * ipp + cus - 1) / cus;
  return ppw * ipp <= rounds_layerwise ? ppw : 0;
}

hipError_t launch_conv3x3_body16w_chain(const ConvParams& p, const ChainArgs& c, int feat, hipStream_t stream, int ablate, bool x3) {
  if (!p.wpk || !p.bias || !c.hi || !c.lo || !c.t || !c.out_f32) return hipErrorInvalidValue;
  if (x3) {
    if (ablate != 0) return hipErrorInvalidValue;
    if (feat == 128) return launch_body16w_chain_one<64, 128, 0, true>(p, c, stream);
    if (feat == 256) return launch_body16w_chain_one<128, 256, 0, true>(p, c, stream);
    return hipErrorInvalidValue;
  }
#ifdef DSEN2_DIAG
  if (ablate == 1024 && feat == 256) return launch_body16w_chain_one<128, 256, 1024>(p, c, stream);
  if (ablate == 1024 && feat == 128) return launch_body16w_chain_one<64, 128, 1024>(p, c, stream);
  if (ablate == 2048 && feat == 256) return launch_body16w_chain_one<128, 256, 2048>(p, c, stream);
  if (ablate == 4096 && feat == 256) return launch_body16w_chain_one<128, 256, 4096>(p, c, stream);
  if (ablate == 5120 && feat == 256) return launch_body16w_chain_one<128, 256, 5120>(p, c, stream);
  if (ablate == 3 && feat == 256) return launch_body16w_chain_one<128, 256, 3>(p, c, stream);
#endif
  if (ablate != 0) return hipErrorInvalidValue;
  if (feat == 128) return launch_body16w_chain_one<64, 128>(p, c, stream);
  if (feat == 256) return launch_body16w_chain_one<128, 256>(p, c, stream);
  return hipErrorInvalidValue;
}

template <int F>
static hipError_t launch_body16w_feat(const ConvParams& p, int epilogue, int ablate, hipStream_t stream, int grid_cap) {
#ifdef DSEN2_DIAG
#define DSEN2_ABL(M)                                                                                         \
  if (ablate == M)                                                                                           \
    return epilogue == kEpiRelu       ? launch_body16w_one<F / 2, F, kEpiRelu, M>(p, stream, grid_cap)                  \
           : epilogue == kEpiResidual ? launch_body16w_one<F / 2, F, kEpiResidual, M>(p, stream, grid_cap)              \
                                      : launch_body16w_one<F / 2, F, kEpiResidualF32, M>(p, stream, grid_cap);
  DSEN2_ABL(1) DSEN2_ABL(2) DSEN2_ABL(3) DSEN2_ABL(4) DSEN2_ABL(8) DSEN2_ABL(12) DSEN2_ABL(15) DSEN2_ABL(16) DSEN2_ABL(31) DSEN2_ABL(32) DSEN2_ABL(96) DSEN2_ABL(128) DSEN2_ABL(256)
#undef DSEN2_ABL
#endif
  if (ablate != 0) return hipErrorInvalidValue;
  if (epilogue == kEpiRelu) return launch_body16w_one<F / 2, F, kEpiRelu>(p, stream, grid_cap);
  if (epilogue == kEpiResidual) return launch_body16w_one<F / 2, F, kEpiResidual>(p, stream, grid_cap);
  if (epilogue == kEpiResidualF32) return launch_body16w_one<F / 2, F, kEpiResidualF32>(p, stream, grid_cap);
  return hipErrorInvalidValue;
}

hipError_t launch_conv3x3_body16w(const ConvParams& p, int feat, int epilogue, int ablate, hipStream_t stream, int grid_cap) {
  if (epilogue != kEpiRelu && (!p.aux || !p.out2)) return hipErrorInvalidValue;
  if (feat == 128) return launch_body16w_feat<128>(p, epilogue, ablate, stream, grid_cap);
  if (feat == 256) return launch_body16w_feat<256>(p, epilogue, ablate, stream, grid_cap);
  return hipErrorInvalidValue;
}

// ---- fp32 channels-last tensor <-> blocked (hi, lo) planes (after the first convolution; test hooks) ----
// One thread per (pixel, 8-channel block).  Threads of a workgroup cover 32 consecutive pixels x all blocks of an
// image, block index fastest on the fp32 side (32-byte pieces of one pixel's 4*C bytes) and pixel index fastest
// on the plane side, through LDS so that both sides are written / read in contiguous runs.
template <bool SPLIT>
__global__ __launch_bounds__(256) void split_join_kernel(float* __restrict__ f32, uint4* __restrict__ hi, uint4* __restrict__ lo,
                                                         int img_pix, int nblk, long long total_groups) {
  // a group = 32 consecutive pixels of one image x all nblk blocks; LDS: [32 px][nblk][8 floats] (+1 float4 pad per pixel)
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int pitch = nblk * 8 + 4;
  const int groups_per_img = (img_pix + 31) / 32;
  for (long long g = blockIdx.x; g < total_groups; g += gridDim.x) {
    const long long img = g / groups_per_img;
    const int p0 = (int)(g - img * groups_per_img) * 32;
    const int npx = min(32, img_pix - p0);
    float* const src = f32 + ((size_t)img * img_pix + p0) * (size_t)(nblk * 8);
    uint4* const hi_img = hi + (size_t)img * nblk * img_pix;
    uint4* const lo_img = lo + (size_t)img * nblk * img_pix;
    if constexpr (SPLIT) {
      for (int i = threadIdx.x; i < npx * nblk * 2; i += blockDim.x) {           // float4 pieces, channel-fastest
        const int px = i / (nblk * 2), q = i - px * (nblk * 2);
        *reinterpret_cast<f32x4*>(tile + px * pitch + q * 4) = *reinterpret_cast<const f32x4*>(src + (size_t)px * nblk * 8 + q * 4);
      }
      __syncthreads();
      for (int i = threadIdx.x; i < npx * nblk; i += blockDim.x) {               // (block, pixel), pixel-fastest
        const int blk = i / npx, px = i - blk * npx;
        const unsigned* v = reinterpret_cast<const unsigned*>(tile + px * pitch + blk * 8);
        unsigned h0, l0, h1, l1, h2, l2, h3, l3;
        split2(v[0], v[1], h0, l0); split2(v[2], v[3], h1, l1); split2(v[4], v[5], h2, l2); split2(v[6], v[7], h3, l3);
        hi_img[(size_t)blk * img_pix + p0 + px] = make_uint4(h0, h1, h2, h3);
        lo_img[(size_t)blk * img_pix + p0 + px] = make_uint4(l0, l1, l2, l3);
      }
      __syncthreads();
    } else {
      for (int i = threadIdx.x; i < npx * nblk; i += blockDim.x) {
        const int blk = i / npx, px = i - blk * npx;
        const uint4 h = hi_img[(size_t)blk * img_pix + p0 + px], l = lo_img[(size_t)blk * img_pix + p0 + px];
        unsigned* v = reinterpret_cast<unsigned*>(tile + px * pitch + blk * 8);
        join2(h.x, l.x, v[0], v[1]); join2(h.y, l.y, v[2], v[3]); join2(h.z, l.z, v[4], v[5]); join2(h.w, l.w, v[6], v[7]);
      }
      __syncthreads();
      for (int i = threadIdx.x; i < npx * nblk * 2; i += blockDim.x) {
        const int px = i / (nblk * 2), q = i - px * (nblk * 2);
        *reinterpret_cast<f32x4*>(src + (size_t)px * nblk * 8 + q * 4) = *reinterpret_cast<const f32x4*>(tile + px * pitch + q * 4);
      }
      __syncthreads();
    }
  }
}

// precision 2: fp32 channels-last tensor -> the residual stream's tensors: hx [n][2][C/8][h][w][8] (plane 0 = hi, the bf16
// rounding (ties away) of the bit pattern; plane 1 = xl = bf16(x - hi), RNE) and lo [n][C/8][h][w][8] (the low halves).
// Same thread layout as split_join_kernel<true>.
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ f32, uint4* __restrict__ hx, uint4* __restrict__ lo,
                                                     int img_pix, int nblk, long long total_groups) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int pitch = nblk * 8 + 4;
  const int groups_per_img = (img_pix + 31) / 32;
  for (long long g = blockIdx.x; g < total_groups; g += gridDim.x) {
    const long long img = g / groups_per_img;
    const int p0 = (int)(g - img * groups_per_img) * 32;
    const int npx = min(32, img_pix - p0);
    const float* const src = f32 + ((size_t)img * img_pix + p0) * (size_t)(nblk * 8);
    uint4* const hi_img = hx + (size_t)img * 2 * nblk * img_pix;
    uint4* const xl_img = hi_img + (size_t)nblk * img_pix;
    uint4* const lo_img = lo + (size_t)img * nblk * img_pix;
    for (int i = threadIdx.x; i < npx * nblk * 2; i += blockDim.x) {           // float4 pieces, channel-fastest
      const int px = i / (nblk * 2), q = i - px * (nblk * 2);
      *reinterpret_cast<f32x4*>(tile + px * pitch + q * 4) = *reinterpret_cast<const f32x4*>(src + (size_t)px * nblk * 8 + q * 4);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < npx * nblk; i += blockDim.x) {               // (block, pixel), pixel-fastest
      const int blk = i / npx, px = i - blk * npx;
      const unsigned* v = reinterpret_cast<const unsigned*>(tile + px * pitch + blk * 8);
      unsigned h[4], l[4], x[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        split2(v[2 * k], v[2 * k + 1], h[k], l[k]);
        x[k] = pack_bf16(__builtin_bit_cast(float, v[2 * k]) - __builtin_bit_cast(float, h[k] << 16),
                         __builtin_bit_cast(float, v[2 * k + 1]) - __builtin_bit_cast(float, h[k] & 0xffff0000u));
      }
      hi_img[(size_t)blk * img_pix + p0 + px] = make_uint4(h[0], h[1], h[2], h[3]);
      xl_img[(size_t)blk * img_pix + p0 + px] = make_uint4(x[0], x[1], x[2], x[3]);
      lo_img[(size_t)blk * img_pix + p0 + px] = make_uint4(l[0], l[1], l[2], l[3]);
    }
    __syncthreads();
  }
}

hipError_t launch_split3_f32(const float* in_nhwc, void* hx, void* lo, int n, int h, int w, int c, hipStream_t stream) {
  if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || c % 8 != 0 || c > 512) return hipErrorInvalidValue;
  const int img_pix = h * w, nblk = c / 8;
  const long long groups = (long long)n * ((img_pix + 31) / 32);
  const size_t lds = (size_t)32 * (nblk * 8 + 4) * sizeof(float);
  const unsigned grid = (unsigned)(groups < 256 * 8 ? groups : 256 * 8);
  static KernelOnce once;
  hipError_t e = once.prepare(reinterpret_cast<const void*>(split3_kernel), (size_t)32 * (512 + 4) * sizeof(float), nullptr);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(split3_kernel, dim3(grid), dim3(256), lds, stream, in_nhwc, reinterpret_cast<uint4*>(hx),
                     reinterpret_cast<uint4*>(lo), img_pix, nblk, groups);
  return hipGetLastError();
}

template <bool SPLIT>
static hipError_t launch_split_join(float* f32, void* hi, void* lo, int n, int h, int w, int c, hipStream_t stream) {
  if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || c % 8 != 0 || c > 512) return hipErrorInvalidValue;
  const int img_pix = h * w, nblk = c / 8;
  const long long groups = (long long)n * ((img_pix + 31) / 32);
  const size_t lds = (size_t)32 * (nblk * 8 + 4) * sizeof(float);
  const unsigned grid = (unsigned)(groups < 256 * 8 ? groups : 256 * 8);
  static KernelOnce once;        // c = 512 needs 66,048 B of dynamic LDS, above the 64 KiB default
  hipError_t e = once.prepare(reinterpret_cast<const void*>(split_join_kernel<SPLIT>), (size_t)32 * (512 + 4) * sizeof(float), nullptr);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(split_join_kernel<SPLIT>, dim3(grid), dim3(256), lds, stream, f32, reinterpret_cast<uint4*>(hi),
                     reinterpret_cast<uint4*>(lo), img_pix, nblk, groups);
  return hipGetLastError();
}

hipError_t launch_split_f32(const float* in_nhwc, void* hi, void* lo, int n, int h, int w, int c, hipStream_t stream) {
  return launch_split_join<true>(const_cast<float*>(in_nhwc), hi, lo, n, h, w, c, stream);
}

hipError_t launch_join_f32(const void* hi, const void* lo, float* out_nhwc, int n, int h, int w, int c, hipStream_t stream) {
  return launch_split_join<false>(out_nhwc, const_cast<void*>(hi), const_cast<void*>(lo), n, h, w, c, stream);
}

}  // namespace dsen2
