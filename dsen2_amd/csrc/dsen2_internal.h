// Internal declarations shared by the HIP translation units of libdsen2_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>

namespace dsen2 {

// ---- data layout ------------------------------------------------------------------------------
// Activations inside the network: NHWC float32, [n][y][x][c], c contiguous (512 B per pixel at F=128).
// Packed conv weights (one buffer per Conv2D):
//   wpk[slab][cc][tap][g][o][j]  float32
//     slab : output-channel slab of NT channels handled by one workgroup      (COUT_PAD / NT)
//     cc   : input-channel chunk of KC channels                               (CIN_PAD / KC)
//     tap  : dy*3 + dx                                                        (9)
//     g    : group of 4 input channels inside the chunk                       (KC / 4)
//     o    : output channel inside the slab                                   (NT)
//     j    : input channel inside the group                                   (4)
//   value = K_hwio[dy][dx][cc*KC + 4g + j][slab*NT + o]   (0 where the index is padding)
// so one (slab, cc, tap) chunk is KC*NT contiguous floats that are copied verbatim into LDS, and one
// ds_read_b128 at [g][o] hands a lane the 4 A-operand values W[o][4g .. 4g+3].

constexpr int kTile = 16;                 // output pixels per workgroup edge (16 x 16 tile)
constexpr int kHalo = kTile + 2;          // 18
constexpr int kHaloPix = kHalo * kHalo;   // 324

enum Epilogue : int {
  kEpiRelu = 0, kEpiResidual = 1, kEpiSkipNCHW = 2,
  kEpiResidualF32 = 3,   // bf16 body kernel only
  kEpiReluSplit = 4      // generic first convolution of a precision-1 model (conv3x3_mfma.hip; band groups conv3x3_first16.hip does
                         // not take): relu(conv + b) written as blocked (hi, lo) planes (out, out2)
};

struct ConvParams {
  const float* in;     // NHWC [n][h][w][CIN_PAD]            (bf16 body kernel: bf16 NHWC)
  const float* wpk;    // packed weights (layout above)
  const float* bias;   // [COUT_PAD]
  const float* aux;    // kEpiResidual: NHWC [n][h][w][COUT]; kEpiSkipNCHW: NCHW [n][cout_real][h][w]
                       // (bf16 body kernel, residual epilogues: the `hi` plane of the residual stream)
  float* out;          // NHWC [n][h][w][COUT] or NCHW [n][cout_real][h][w]  (bf16 body kernel, kEpiRelu: bf16 NHWC)
  void* out2;          // bf16 body kernel, residual epilogues: the `lo` plane of the residual stream; otherwise unused
  int n, h, w;
  int tiles_x, tiles_y;
  int cout_real;       // kEpiSkipNCHW only
  float res_scale;     // residual epilogues only
  unsigned long long* diag;   // DSEN2_DIAG builds, ablation bit 32: in-kernel time stamps (tools/stamp_body_conv.py); else NULL
};

// Kernel-structure choices of a model.  The product library always uses the defaults; the diagnostic build
// (-DDSEN2_DIAG, tools/ only) can change them through dsen2_diag_set for A/B measurements.
struct Tuning {
  int body_variant = 14;   // fp32 F->F body convolution: 11-14 = conv3x3_body32.hip sub-variants 0-3; 0 = one tile per
                           // workgroup (conv3x3_mfma.hip, the independent first implementation)
  int out_variant = 2;     // last layer, Cout <= 8: 2 = the tap-expanded matrix-core kernel (conv3x3_out_mfma.hip) where the
                           // shape fits it, else the vector-unit kernel (conv3x3_out.hip); 3 = always the vector-unit kernel;
                           // 0 (and Cout > 8) = padded 32-wide MFMA block (conv3x3_mfma.hip, the reference structure)
  int ablate = 0;          // timing-only ablation mask of the persistent body kernels (DSEN2_DIAG builds; wrong outputs)
  int grid_cap = 0;        // DSEN2_DIAG builds: launch at most this many workgroups of the bf16 body kernel (0 = one per CU)
  int first_ablate = 0;    // DSEN2_DIAG builds: timing-only ablation mask of the first convolution (conv3x3_first.hip)
  int out_ablate = 0;      // DSEN2_DIAG builds: timing-only ablation mask of conv3x3_out_mfma.hip
  int chain = 1;           // precision 1: one persistent launch over all body layers when the batch gives every CU whole patches
};

// Per-kernel launch preparation: the dynamic-LDS attribute is a property of (kernel, device) and is set once per pair,
// under a mutex — host threads driving different devices (or one device from several streams) may reach a launcher
// at the same time.  One function-local static instance per kernel instantiation (its construction is thread-safe).
struct KernelOnce {
  std::mutex mu;
  bool done[64] = {};
  int cus[64] = {};
  // current device -> *dev_cus (its CU count); sets MaxDynamicSharedMemorySize of `kern` there on first use
  hipError_t prepare(const void* kern, size_t lds_bytes, int* dev_cus) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mu);
    if (!done[dev]) {
      if (lds_bytes > 0) {
        e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
      }
      e = hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev);
      if (e != hipSuccess) return e;
      done[dev] = true;
    }
    if (dev_cus) *dev_cus = cus[dev];
    return hipSuccess;
  }
};

// Supported (CIN_PAD, COUT_PAD, epilogue) combinations; returns hipErrorInvalidValue otherwise.
struct PackGeom { int kc, nt, cin_pad, cout_pad, variant; };
hipError_t launch_conv3x3(const ConvParams& p, const PackGeom& geom, int epilogue, int ablate, hipStream_t stream);
// last layer, F -> Cout <= 8, on the vector units (conv3x3_out.hip; PackGeom variant 8, weights packed by
// pack_out_valu_weights_host)
hipError_t launch_conv3x3_out_valu(const ConvParams& p, int feat, hipStream_t stream);
void pack_out_valu_weights_host(const float* kernel_hwio, int cin, int cout, float* dst);
// last layer, F -> Cout <= 6, the nine taps expanded into the M side of one GEMM (conv3x3_out_mfma.hip).  Its weights follow
// the vector-unit kernel's in a variant-8 buffer (9 * cin * 8 floats further).  *taken = false: shape not supported,
// nothing launched.
hipError_t launch_conv3x3_out_mfma(const ConvParams& p, int feat, hipStream_t stream, bool* taken, int ablate = 0);
size_t out_mfma_weight_floats(int cin);
void pack_out_mfma_weights_host(const float* kernel_hwio, int cin, int cout, float* dst);
// First convolution reading the NCHW inputs directly (conv3x3_first.hip, exact fp32): p.in = x10, p.aux = x20; weights packed
// with PackGeom{16, 128, 16, cout, .}; epilogue kEpiRelu only (p.out fp32 NHWC).
// hipErrorNotSupported for channel counts other than 10 / 12 (then: launch_pack_inputs + launch_conv3x3).
struct FirstInputs { const float* x60; int c10, c20, c60; };
hipError_t launch_conv3x3_first(const ConvParams& p, const FirstInputs& f, int cout, int epilogue, hipStream_t stream, int ablate = 0);
// ... of a precision-1 / -2 model, on the bf16 matrix cores (conv3x3_first16.hip): same inputs; p.wpk = the buffer
// pack_first16_weights_host fills (first16_weight_u16(cout, x3) uint16); x3 = false: p.out / p.out2 = the blocked (hi, lo)
// planes of the residual stream (bf16 operands: out = relu(bf16(x) * bf16(w) + b)); x3 = true: p.out = hx (hi | xl planes),
// p.out2 = lo16 — launch_split3_f32's tensors — with three-MFMA products (xh*wh + xh*wl + xl*wh).  hipErrorNotSupported as above.
hipError_t launch_conv3x3_first16(const ConvParams& p, const FirstInputs& f, int cout, bool x3, hipStream_t stream);
size_t first16_weight_u16(int cout, bool x3);
void pack_first16_weights_host(const float* kernel_hwio, int cin, int cout, bool x3, uint16_t* dst);
// DMA-fed fp32 kernel (conv3x3_body32.hip): F = 128 or 256, images < 2 GiB; weights packed with KC=32, NT=128
bool body32_supports(const ConvParams& p, int cout);
hipError_t launch_conv3x3_body32(const ConvParams& p, int feat, int epilogue, int sub, int ablate, hipStream_t stream);
// kernel HWIO fp32 -> bf16 packed [slab][cc (chunk_ch channels)][step][g (groups of 8 ch)][o(128)][8]; dst holds
// 9*cin*cout uint16.  step = tap, except with perm16 (conv3x3_body16w.hip walks the taps dx-major: step s carries tap
// (dy, dx) = (s % 3, s / 3)).  perm16: row o of a slab holds output channel 32*(o>>5) + 8*((o&15)>>2) + 4*((o>>4)&1) + (o&3),
// so that the two 16-row accumulators of a 32-channel pair give a lane 8 consecutive channels (conv3x3_body16w.hip)
void pack_conv_weights_bf16_host(const float* kernel_hwio, int cin, int cout, int chunk_ch, bool perm16, uint16_t* dst);
// bf16-operand body convolution, wide tile (conv3x3_body16w.hip), F = 128 or 256.  16-bit tensors are BLOCKED:
// [n][C/8][h][w][8] (an 8-channel block = a plane of 16-byte pixels).  p.in bf16 blocked; weights packed by
// pack_conv_weights_bf16_host(chunk_ch = 32, perm16).  kEpiRelu: p.out bf16 blocked.  kEpiResidual: the residual
// stream as two blocked 16-bit tensors p.aux (hi = bf16 rounding, the next operand) / p.out2 (lo), updated in place.
// kEpiResidualF32: same inputs, result to p.out as fp32 NHWC (last block).  `ablate` != 0 only in DSEN2_DIAG builds.
hipError_t launch_conv3x3_body16w(const ConvParams& p, int feat, int epilogue, int ablate, hipStream_t stream, int grid_cap = 0);
// precision 2 ("bf16x3", conv3x3_body16w.hip X3): fp32-grade products on the bf16 matrix cores — every operand is two bf16
// numbers (hi + lo), a product is hi*hi + hi*lo + lo*hi.  16-bit OPERAND tensors carry two planes per image,
// [n][2][F/8][h][w][8] (hi | lo).  p.in = such a tensor; weights packed by pack_conv_weights_bf16x3_host.
//   kEpiRelu: p.out = relu(conv + bias) as a two-plane tensor.
//   kEpiResidual: the residual stream = p.aux (two planes: hi = bf16 rounding (ties away) of the bit pattern, xl = bf16(x - hi))
//     + p.out2 (lo16: the low halves, one plane): (hi, lo16) hold the exact fp32 value as in precision 1; updated in place.
//   kEpiResidualF32: same inputs, result to p.out as fp32 NHWC.
hipError_t launch_conv3x3_body16w_x3(const ConvParams& p, int feat, int epilogue, hipStream_t stream);
// kernel HWIO fp32 (3,3,cin,cout) -> the packed bf16 layout of a (3,3,3*cin,cout) convolution whose input chunk 3*cc + j holds
// (wh, wl, wh)[j] of real chunk cc (wh = RNE bf16 of w, wl = RNE bf16 of w - wh); dst holds 27*cin*cout uint16
void pack_conv_weights_bf16x3_host(const float* kernel_hwio, int cin, int cout, uint16_t* dst);
// fp32 NHWC -> the precision-2 residual stream: hx [n][2][C/8][h][w][8] (hi | xl), lo [n][C/8][h][w][8] (c % 8 == 0)
hipError_t launch_split3_f32(const float* in_nhwc, void* hx, void* lo, int n, int h, int w, int c, hipStream_t stream);
// One launch over all 2d residual-block convolutions of a precision-1 network (conv3x3_body16w.hip, CHAIN): a workgroup
// owns whole patches through every layer, so layers need no cross-workgroup synchronisation.
struct ChainArgs {
  void* hi;                   // residual stream: bf16 rounding plane (the convolutions' operand) ...
  void* lo;                   // ... and low halves
  void* t;                    // relu(conv-A), bf16 blocked
  float* out_f32;             // last block's output, fp32 NHWC
  unsigned layer_stride;      // bytes between the packed weights (and between the biases) of consecutive body layers
  int n_layers;               // 2 * d
  int patches_per_wg;
  int seamless;               // layer boundaries without a drain (conv3x3_body16w.hip; filled in by the launcher)
};
// p.wpk / p.bias = the first body layer's packed weights / bias (the following layers' lie layer_stride bytes further
// each); p.n, p.h, p.w, p.res_scale as usual; the tensors come from `c`.  c.patches_per_wg is filled in here.
// x3: the precision-2 form (c.hi = the stream's two-plane operand tensor hi | xl, c.t two planes, weights packed by
// pack_conv_weights_bf16x3_host)
hipError_t launch_conv3x3_body16w_chain(const ConvParams& p, const ChainArgs& c, int feat, hipStream_t stream, int ablate = 0,
                                        bool x3 = false);
// > 0: the chain kernel keeps every CU as busy as the per-layer launches do for this batch (that many patches per
// workgroup); 0: use the per-layer kernels
int body16w_chain_patches_per_wg(int n, int h, int w, int feat, int cus);
// fp32 NHWC tensor <-> blocked (hi, lo) tensors: hi = (u + 0x8000) >> 16, lo = u & 0xffff per value (c % 8 == 0)
hipError_t launch_split_f32(const float* in_nhwc, void* hi, void* lo, int n, int h, int w, int c, hipStream_t stream);
hipError_t launch_join_f32(const void* hi, const void* lo, float* out_nhwc, int n, int h, int w, int c, hipStream_t stream);
// Geometry helpers for packing
bool conv_pack_geometry(int cin, int cout, int epilogue, const Tuning& tune, PackGeom* g);
size_t packed_weight_floats(const PackGeom& g);
// host_kernel HWIO (3,3,cin,cout) -> packed layout (host memory, zero padded)
void pack_conv_weights_host(const float* kernel_hwio, int cin, int cout, const PackGeom& g, float* dst);

// ---- elementwise / gather kernels (patch_ops.hip) ---------------------------------------------
// concat(x10,x20,x60) NCHW -> NHWC with 16 channels (zero padded): folds keras Concatenate(axis=1).
hipError_t launch_pack_inputs(const float* x10, const float* x20, const float* x60, int c10, int c20, int c60,
                              float* out_nhwc16, int n, int h, int w, hipStream_t stream);
// general = true: always the general kernel (the windowed one otherwise takes up-sampling by 2 or more)
hipError_t launch_upsample(const float* in, float* out, int planes, int h, int w, int oh, int ow, float post_div,
                           hipStream_t stream, bool general = false);
hipError_t launch_tile_gather(const float* img, int H, int W, int C, int border, const int* origins, int count,
                              int P, float divisor, float* patches, hipStream_t stream);
// rows [row0, row1) of the image (0, H = all of it)
hipError_t launch_recompose(const float* patches, int count, int C, int P, int border, float* img, int H, int W,
                            float scale, int row0, int row1, hipStream_t stream);

}  // namespace dsen2
