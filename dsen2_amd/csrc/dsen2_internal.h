// Internal declarations shared by the HIP translation units of libdsen2_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

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

enum Epilogue : int { kEpiRelu = 0, kEpiResidual = 1, kEpiSkipNCHW = 2 };

struct ConvParams {
  const float* in;     // NHWC [n][h][w][CIN_PAD]
  const float* wpk;    // packed weights (layout above)
  const float* bias;   // [COUT_PAD]
  const float* aux;    // kEpiResidual: NHWC [n][h][w][COUT]; kEpiSkipNCHW: NCHW [n][cout_real][h][w]
  float* out;          // NHWC [n][h][w][COUT] or NCHW [n][cout_real][h][w]  (bf16 body kernel, kEpiRelu: bf16 NHWC)
  void* out2;          // bf16 body kernel, kEpiResidual: bf16 NHWC copy of `out`; otherwise unused
  int n, h, w;
  int tiles_x, tiles_y;
  int cout_real;       // kEpiSkipNCHW only
  float res_scale;     // kEpiResidual only
  int stagger;         // persistent body kernel: start-delay quantum (x 8128 cycles) per (workgroup mod 4); 0 = off
};

// Supported (CIN_PAD, COUT_PAD, epilogue) combinations; returns hipErrorInvalidValue otherwise.
struct PackGeom { int kc, nt, cin_pad, cout_pad, variant; };
hipError_t launch_conv3x3(const ConvParams& p, const PackGeom& geom, int epilogue, hipStream_t stream);
extern int g_body_variant;
extern int g_out_variant;
// last layer, F -> Cout<=16, 16x16x4 MFMA (conv3x3_out.hip); weights packed with KC=16, NT=16
hipError_t launch_conv3x3_out(const ConvParams& p, int feat, hipStream_t stream);
extern int g_body_stagger;  // tuning key 3
extern int g_body_ablate;   // timing-only ablation mask of the persistent body kernel (0 = off)
// deferred-epilogue persistent kernel (conv3x3_bodyd.hip): fp32 F=128 and bf16 F=256 only, tensors < 4 GiB
bool bodyd_supports(const ConvParams& p, int cout);
hipError_t launch_conv3x3_bodyd(const ConvParams& p, int feat, int epilogue, bool bf16, hipStream_t stream);
// DMA-fed fp32 kernel (conv3x3_body32.hip): F = 128 or 256, images < 2 GiB; weights packed with KC=32, NT=128
bool body32_supports(const ConvParams& p, int cout);
hipError_t launch_conv3x3_body32(const ConvParams& p, int feat, int epilogue, int sub, hipStream_t stream);
// bf16-operand form of the persistent kernel: in bf16 NHWC, weights packed by pack_conv_weights_bf16_host
hipError_t launch_conv3x3_body_bf16(const ConvParams& p, int feat, int epilogue, int variant, hipStream_t stream);
// kernel HWIO fp32 -> bf16 packed [slab][cc(64 ch)][tap][g(8 groups of 8 ch)][o(128)][8]; dst holds 9*cin*cout uint16
// perm16: row o of a slab holds output channel 32*(o>>5) + 8*((o&15)>>2) + 4*((o>>4)&1) + (o&3) (conv3x3_body16.hip)
void pack_conv_weights_bf16_host(const float* kernel_hwio, int cin, int cout, int chunk_ch, bool perm16, uint16_t* dst);
extern int g_bf16_variant;   // tuning key 4 (read when weights are packed and when the kernel is launched)
inline int bf16_chunk_channels(int) { return 64; }   // every structure stages 64-channel chunks
inline bool bf16_perm16(int variant) { return variant >= 4 && variant <= 7; }
// 16x16x32-MFMA form fed by LDS-DMA (conv3x3_body16.hip), F = 256; weights packed with perm16
hipError_t launch_conv3x3_body16(const ConvParams& p, int feat, int epilogue, int sub, hipStream_t stream);
hipError_t launch_f32_to_bf16(const float* in, void* out_bf16, size_t count, hipStream_t stream);
// persistent pipelined F->F kernel (conv3x3_body.hip); weights packed with KC=32, NT=128
hipError_t launch_conv3x3_body(const ConvParams& p, int feat, int epilogue, int variant, hipStream_t stream);
// Geometry helpers for packing
bool conv_pack_geometry(int cin, int cout, int epilogue, PackGeom* g);
size_t packed_weight_floats(const PackGeom& g);
// host_kernel HWIO (3,3,cin,cout) -> packed layout (host memory, zero padded)
void pack_conv_weights_host(const float* kernel_hwio, int cin, int cout, const PackGeom& g, float* dst);

// ---- elementwise / gather kernels (patch_ops.hip) ---------------------------------------------
// concat(x10,x20,x60) NCHW -> NHWC with 16 channels (zero padded): folds keras Concatenate(axis=1).
hipError_t launch_pack_inputs(const float* x10, const float* x20, const float* x60, int c10, int c20, int c60,
                              float* out_nhwc16, int n, int h, int w, hipStream_t stream);
hipError_t launch_upsample(const float* in, float* out, int planes, int h, int w, int oh, int ow, float post_div,
                           hipStream_t stream);
hipError_t launch_tile_gather(const float* img, int H, int W, int C, int border, const int* origins, int count,
                              int P, float divisor, float* patches, hipStream_t stream);
hipError_t launch_recompose(const float* patches, int count, int C, int P, int border, float* img, int H, int W,
                            float scale, hipStream_t stream);

}  // namespace dsen2
