/*
 * dsen2_hip.h — C ABI of libdsen2_hip.so: the MI355X (gfx950) DSen2 / VDSen2 inference path.
 *
 * The reference (ACMEAtronOmatic/DSen2) is pure Python and has no FFI layer; the "operator API"
 * below the drop-in boundary is keras' Model object.  Each entry point names the reference
 * interface it stands in for (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain C types only; every `const float* dev_*` / `float* dev_*` is a DEVICE pointer owned by
 *     the caller (e.g. a PyTorch-ROCm tensor's data_ptr()); `host_*` pointers are host memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All work is enqueued on
 *     it; nothing synchronises unless stated.
 *   - every function returns DSEN2_OK (0) or a negative error code and never throws;
 *     dsen2_last_error() returns a thread-local description of the last failure.
 *   - the library uses the calling thread's current HIP device; one model handle per device: a handle belongs to the
 *     device that was current when it was created, and every call that takes a handle returns DSEN2_ERR_INVALID when the
 *     calling thread's current device is another one (its weights live on the handle's device).
 *   - no C++ exception crosses the ABI and no path of the library calls abort(): host-side failures (out of memory
 *     included) come back as an error code.  (A fault raised by the GPU itself is the HIP runtime's to report.)
 *   - activations handed across the ABI are NCHW float32 (the reference runs keras in
 *     'channels_first', utils/DSen2Net.py:6); NHWC is internal.
 */
#ifndef DSEN2_HIP_H
#define DSEN2_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSEN2_OK 0
#define DSEN2_ERR_INVALID (-1)     /* bad argument / unsupported shape */
#define DSEN2_ERR_HIP (-2)         /* a HIP runtime call failed */
#define DSEN2_ERR_NO_WEIGHTS (-3)  /* forward before load_weights */
#define DSEN2_ERR_WORKSPACE (-4)   /* workspace too small */
#define DSEN2_ERR_NO_DEVICE (-5)   /* no gfx950 device visible */
#define DSEN2_ERR_NOMEM (-6)       /* the HOST ran out of memory inside the library (std::bad_alloc) — not a caller error */
#define DSEN2_ERR_INTERNAL (-7)    /* any other C++ exception caught at the boundary (a library bug, never a caller error) */
/* Shape limit of every entry point that takes (n, h, w): ONE image's activation tensor must stay below 2^31 bytes
 * (h * w * feature_size < 2^29 values) — the kernels address an image through a 32-bit buffer descriptor.  Checked up
 * front (DSEN2_ERR_INVALID before anything is enqueued), the same for all three precisions; there is no limit on n. */

typedef struct dsen2_model dsen2_model;

const char *dsen2_version(void);
const char *dsen2_last_error(void);
/* Number of visible HIP devices whose gcnArchName starts with "gfx950"; <0 on error. */
int dsen2_device_count(void);

/* ---- network object -------------------------------------------------------------------------
 * dsen2_model_create  <->  s2model(input_shape, num_layers, feature_size)   utils/DSen2Net.py:18-43
 *   c10/c20/c60: channel counts of the 10 m / 20 m / 60 m inputs (c60 = 0 for the 2-input net).
 *   The output has the channel count of the last input and that input is added back (:35-41).
 *   feature_size must be a multiple of 128 (reference uses 128 and 256, testing/supres.py:56,59).
 *   precision: 0 = fp32 everywhere (exact-f32 MFMA); 1 = bf16 operands for the residual-block convolutions
 *   (v_mfma_f32_16x16x32_bf16), fp32 accumulation, an exact fp32 residual stream (kept as two 16-bit planes, see
 *   dsen2_split_f32), the FIRST convolution in the same arithmetic (bf16 operands, fp32 accumulate: dsen2_conv3x3_first_planes;
 *   band groups other than 4 + 6 (+ 2): fp32), fp32 last convolution; 2 = "bf16x3": the first and the residual-block convolutions on the bf16 matrix
 *   cores with every fp32 operand split into two bf16 numbers (x = hi + lo, 16 significant bits) and a product taken as
 *   hi*hi + hi*lo + lo*hi — three MFMAs at 16 x the fp32 MFMA rate, fp32 accumulation, the same exact fp32 residual
 *   stream; whole-network error ~1e-5 in the normalised domain (fp32: 3e-7, precision 1: 4e-3), inside the 1e-4 gate.
 *   Opt-in: it is not the reference's arithmetic (keras computes in fp32) and never the headline benchmark's.
 *   A model's kernel structures are fixed when it is created; the only process-global state of the library are
 *   per-(kernel, device) launch attributes, set once under a mutex — any number of handles (one per rank / device)
 *   coexist, and host threads may drive different devices, or one handle from several streams (each call with
 *   its own workspace), concurrently.
 */
int dsen2_model_create(dsen2_model **out, int c10, int c20, int c60, int num_layers, int feature_size,
                       int precision);
void dsen2_model_destroy(dsen2_model *m);

/* Number of float32 parameters in keras order (kernels HWIO (3,3,Cin,Cout) then bias, per Conv2D,
 * in graph order: conv_in, d x (convA, convB), conv_out). */
size_t dsen2_model_num_params(const dsen2_model *m);

/* dsen2_model_load_weights  <->  model.load_weights(predict_file)        testing/supres.py:63
 *   host_flat: `count` float32 in the order above (what a keras-HDF5 -> flat converter emits).
 *   Packs into the MFMA operand layout and uploads; synchronises the device once. */
int dsen2_model_load_weights(dsen2_model *m, const float *host_flat, size_t count);

/* Scratch the caller must provide to dsen2_model_forward for a batch of n patches of h x w. */
int dsen2_model_workspace_bytes(const dsen2_model *m, int n, int h, int w, size_t *bytes);

/* dsen2_model_forward  <->  model.predict([p10, p20(, p60)])             testing/supres.py:65
 *   dev_x10 [n,c10,h,w], dev_x20 [n,c20,h,w], dev_x60 [n,c60,h,w] or NULL, dev_out [n,cout,h,w];
 *   all NCHW float32 device pointers, all at the same (already up-sampled) h x w. */
int dsen2_model_forward(dsen2_model *m, const float *dev_x10, const float *dev_x20, const float *dev_x60,
                        float *dev_out, int n, int h, int w, void *dev_workspace, size_t workspace_bytes,
                        void *stream);

/* How many kernel launches the 2*num_layers residual-block convolutions of one forward of n patches of h x w take on
 * the current device: 2*num_layers (one per convolution), or 1 — a precision-1 or -2 model runs them as ONE persistent
 * "chain" launch when the batch gives every CU whole patches (each workgroup then owns its patches through all layers;
 * e.g. 256 patches of 32x32 on 256 CUs) and the patches are at most 64 x 64 (beyond that the per-layer launches are
 * measured faster).  Same results either way, bit for bit.  <0 on error. */
int dsen2_model_body_launches(const dsen2_model *m, int n, int h, int w);

/* Measurement hook (no reference counterpart): `iters` forward passes exactly as dsen2_model_forward enqueues them,
 * with a HIP event recorded on `stream` before the first and after the last residual-block convolution of each
 * pass.  *body_ms_per_launch = that interval / (2*num_layers): the mean duration of ONE of those convolutions inside
 * the running network, whether they are 2*num_layers launches or one chain launch (bench.py's roofline figure).
 * Synchronises the stream. */
int dsen2_model_forward_timed(dsen2_model *m, const float *x10, const float *x20, const float *x60, float *out,
                              int n, int h, int w, void *workspace, size_t workspace_bytes, void *stream, int iters,
                              float *body_ms_per_launch);

/* Measurement hook (no reference counterpart): `warm` plain forward passes, then — without a synchronisation in between,
 * so that the GPU never idles and is at its steady clock — `iters` forward passes exactly as dsen2_model_forward enqueues
 * them, each with FOUR HIP events recorded on `stream`: before the first convolution, before the first and after the last
 * residual-block convolution, after the output convolution.  ms5[0] = mean whole forward, ms5[1] = first convolution,
 * ms5[2] = all residual-block convolutions together, ms5[3] = output convolution — consecutive intervals between the same
 * time stamps, so ms5[1] + ms5[2] + ms5[3] = ms5[0]; ms5[4] = mean time per instrumented pass from the first pass's first
 * event to the last pass's last event (what a pass costs WITH its event records and the gap to the next pass).
 * bench.py's `roofline` object is built from these.  Synchronises the stream at the end. */
int dsen2_model_forward_profile(dsen2_model *m, const float *x10, const float *x20, const float *x60, float *out,
                                int n, int h, int w, void *workspace, size_t workspace_bytes, void *stream, int warm,
                                int iters, float *ms5);

/* ---- single-layer entry points (kernel-level parity tests and benchmarks) -------------------
 * One 3x3 'same' convolution (keras Conv2D as used at utils/DSen2Net.py:10,12,29,35) on NHWC
 * float32 device tensors.  host_kernel is HWIO (3,3,cin,cout), host_bias is [cout].
 *   epilogue 0: out = relu(conv + bias)                         DSen2Net.py:10-11 / :29
 *   epilogue 1: out = dev_aux + res_scale * (conv + bias)       DSen2Net.py:12-15   (aux NHWC [n,h,w,cout])
 *   epilogue 2: out = conv + bias + dev_aux, NCHW out and aux   DSen2Net.py:35,38,41 (aux/out [n,cout,h,w])
 * cin must be a multiple of 16 (zero-pad channels), cout a multiple of 128 for epilogues 0/1 and
 * <= 32 for epilogue 2.  Packs the weights on every call (test path, not the hot path). */
int dsen2_conv3x3_nhwc(const float *dev_in, const float *host_kernel, const float *host_bias,
                       const float *dev_aux, float *dev_out, int n, int h, int w, int cin, int cout,
                       int epilogue, float res_scale, void *stream);

/* The same convolution on the one-tile-per-workgroup kernel (the library's first, register-staged structure):
 * an independent implementation of the arithmetic for cross-checks of the persistent DMA-fed kernels
 * (tests/test_gpu_conv.py, tools/stress_body_conv.py).  Bit-identical results are expected. */
int dsen2_conv3x3_nhwc_ref(const float *dev_in, const float *host_kernel, const float *host_bias,
                           const float *dev_aux, float *dev_out, int n, int h, int w, int cin, int cout,
                           int epilogue, float res_scale, void *stream);

/* 16-bit tensors of a precision-1 model are BLOCKED: [n][C/8][h][w][8] — an 8-channel block is a plane of
 * 16-byte pixels (the layout in which the bf16 kernel's loads, stores and LDS-DMA reads are contiguous runs).
 *
 * The residual stream of such a model: each fp32 value u (its bit pattern) is held in two blocked 16-bit tensors,
 *   hi = (u + 0x8000) >> 16   the bf16 rounding of u (ties away from zero) = the next convolution's operand
 *   lo = u & 0xffff
 * and u = ((hi - (lo >> 15)) << 16) | lo restores it bit for bit (all arithmetic mod 2^16 / 2^32; every bit
 * pattern, NaNs included, round-trips).  dsen2_split_f32: fp32 NHWC [n,h,w,c] -> blocked hi, lo (n*h*w*c uint16
 * each); dsen2_join_f32 is the inverse.  c % 8 == 0, c <= 512. */
int dsen2_split_f32(const float *dev_in_nhwc, void *dev_hi, void *dev_lo, int n, int h, int w, int c, void *stream);
int dsen2_join_f32(const void *dev_hi, const void *dev_lo, float *dev_out_nhwc, int n, int h, int w, int c, void *stream);

/* bf16-operand form of one residual-block convolution (feat -> feat, feat = 128 or 256): dev_in_bf16 is a BLOCKED
 * bf16 tensor; the fp32 HWIO kernel is rounded to bf16 (RNE) while packing; accumulation is fp32, starting from the bias.
 *   epilogue 0: dev_out (bf16 blocked, RNE) = relu(conv + bias)                           DSen2Net.py:10-11
 *   epilogue 1: (dev_res_hi, dev_res_lo) = split(join(hi, lo) + res_scale * (conv + bias)), in place; dev_out unused
 *   epilogue 3: dev_out (fp32 NHWC) = join(hi, lo) + res_scale * (conv + bias)            DSen2Net.py:12-15
 * Test path (packs on every call, synchronises). */
int dsen2_conv3x3_body_bf16(const void *dev_in_bf16, const float *host_kernel, const float *host_bias,
                            void *dev_res_hi, void *dev_res_lo, void *dev_out, int n, int h, int w, int feat,
                            int epilogue, float res_scale, void *stream);

/* precision 2 ("bf16x3") at kernel level.  16-bit OPERAND tensors carry two blocked planes per image:
 * [n][2][C/8][h][w][8], plane 0 = hi (bf16), plane 1 = lo (bf16), value ~ hi + lo.
 * dsen2_split3_f32: fp32 NHWC [n,h,w,c] -> the residual stream of a precision-2 model: dev_hx (two planes: hi = the bf16
 *   rounding of the bit pattern, ties away, as in dsen2_split_f32; xl = bf16(x - hi), round to nearest even) and dev_lo
 *   ([n][c/8][h][w][8], the low halves: (hi, lo) restore x bit for bit with dsen2_join_f32 applied to plane 0).
 * dsen2_conv3x3_body_bf16x3: one residual-block convolution, feat -> feat; the fp32 HWIO kernel is split into (wh, wl) while
 *   packing; dev_in_planes is a two-plane operand tensor.
 *   epilogue 0: dev_out (two planes, RNE both) = relu(conv + bias)
 *   epilogue 1: the stream (dev_res_hx planes hi | xl, dev_res_lo) <- split3(join(hi, lo) + res_scale * (conv + bias)), in place
 *   epilogue 3: dev_out (fp32 NHWC) = join(hi, lo) + res_scale * (conv + bias)
 * Test path (packs on every call, synchronises). */
int dsen2_split3_f32(const float *dev_in_nhwc, void *dev_hx, void *dev_lo, int n, int h, int w, int c, void *stream);
int dsen2_conv3x3_body_bf16x3(const void *dev_in_planes, const float *host_kernel, const float *host_bias,
                              void *dev_res_hx, void *dev_res_lo, void *dev_out, int n, int h, int w, int feat,
                              int epilogue, float res_scale, void *stream);

/* The FIRST convolution of a precision-1 / -2 model at kernel level (utils/DSen2Net.py:24-29: Concatenate(axis=1) of the
 * NCHW inputs + Conv2D(feat, 3x3, 'same') + bias + ReLU) on the bf16 matrix cores, writing the residual stream in the form the
 * residual-block kernels read.  Band groups 4 + 6 (+ 2) only; host_kernel HWIO (3, 3, c10 + c20 + c60, feat).
 *   precision 1: out = relu(sum bf16(x) * bf16(w) + bias) (RNE roundings, fp32 accumulate);
 *                dev_out / dev_out2 = its blocked (hi, lo) planes — dsen2_split_f32's tensors
 *   precision 2: x = xh + xl, w = wh + wl (bf16 each), products xh*wh + xh*wl + xl*wh in fp32;
 *                dev_out = hx (two planes per image: hi | xl), dev_out2 = lo16 — dsen2_split3_f32's tensors
 * Test path (packs on every call, synchronises). */
int dsen2_conv3x3_first_planes(const float *dev_x10, const float *dev_x20, const float *dev_x60, int c10, int c20, int c60,
                               const float *host_kernel, const float *host_bias, int feat, int precision,
                               void *dev_out, void *dev_out2, int n, int h, int w, void *stream);

/* Body-convolution micro-benchmark hook: runs `iters` launches of the 128->128 (or F->F) kernel on
 * caller-provided NHWC buffers with already-packed weights held by `m` (layer index `layer`, 1-based
 * body conv number), after 24 untimed launches of the same kernel (the chip's clock after an idle stretch), and reports the
 * mean kernel time in milliseconds measured with HIP events on `stream`.  Used by bench.py (each epilogue alone on dense
 * random operands, beside the in-network figure).
 * precision-1 models: dev_in is bf16 blocked; odd layers write bf16 blocked to dev_out; even (residual) layers take
 * dev_aux = one fp32-sized buffer holding the blocked hi tensor followed by the lo tensor and update it in place
 * (the last block's residual layer writes fp32 NHWC to dev_out instead). */
int dsen2_model_time_body_conv(dsen2_model *m, int layer, const float *dev_in, const float *dev_aux,
                               float *dev_out, int n, int h, int w, int iters, void *stream,
                               float *ms_per_launch);

/* ---- tiling / up-sampling / recomposition (utils/patches.py) --------------------------------
 * dsen2_upsample_mirror_bilinear  <->  interp_patches            utils/patches.py:11-16
 *   planes x [h,w] -> planes x [oh,ow]; half-pixel-centre bilinear with mirror boundary (skimage
 *   resize mode='reflect'), including the /30000 .. *30000 round trip — scikit-image 0.18.3's float32 arithmetic operation
 *   by operation: the reference's outputs bit for bit.  The result is then divided by
 *   `post_divisor` (1.0 = exact no-op, 2000 folds `p20 /= SCALE`, testing/supres.py:24). */
int dsen2_upsample_mirror_bilinear(const float *dev_in, float *dev_out, int planes, int h, int w, int oh,
                                   int ow, float post_divisor, void *stream);
/* The same on the general kernel (any scale; samples fetched on demand) whatever the scale — the windowed kernel that
 * up-sampling by 2 or more normally takes must give its bits (kernel-level cross-check, like dsen2_conv3x3_nhwc_ref). */
int dsen2_upsample_mirror_bilinear_ref(const float *dev_in, float *dev_out, int planes, int h, int w, int oh,
                                       int ow, float post_divisor, void *stream);

/* dsen2_tile_gather  <->  the pad + crop loops of get_test_patches{,60}   patches.py:27-28,58-72 / :93-95,127-143
 *   dev_img: one HWC float32 image [H,W,C] (unpadded); writes patches [count,C,P,P] NCHW where patch k
 *   has its origin at (dev_origins[2k], dev_origins[2k+1]) in PADDED coordinates (np.pad mode
 *   'symmetric' by `border`, materialised on the fly).  Values are divided by `divisor`
 *   (1.0 = exact copy, 2000 folds `p10 /= SCALE`, testing/supres.py:23; an IEEE float32 divide, so the
 *   result is bit-identical to numpy's). */
int dsen2_tile_gather(const float *dev_img, int H, int W, int C, int border, const int *dev_origins,
                      int count, int P, float divisor, float *dev_patches, void *stream);

/* dsen2_recompose  <->  recompose_images                          utils/patches.py:374-405
 *   dev_patches [count,C,P,P] NCHW (the "a.shape[0] == 1 -> return a[0] uncropped" quirk of
 *   patches.py:375-376 is a host-side transpose, not this function); writes the HWC image [H,W,C].  Tile grid as the
 *   reference: inner = P - 2*border, x_tiles = ceil(W/inner), y_tiles = ceil(H/inner), row-major patch
 *   order, last row/column origin clamped to size - inner.  The reference's sequential loop lets later
 *   patches overwrite earlier ones where the clamped tile overlaps its neighbour; here every output
 *   pixel reads the LAST patch that covers it, which is the same image without a write race.
 *   Values are multiplied by `scale` (1.0, or 2000 to fold testing/supres.py:29).
 *   Requires count >= x_tiles*y_tiles, H >= inner, W >= inner. */
int dsen2_recompose(const float *dev_patches, int count, int C, int P, int border, float *dev_img, int H,
                    int W, float scale, void *stream);
/* The same for rows [row0, row1) of the image only: the rows a caller can finish (and start downloading) while later patches
 * are still being computed.  Only the patches those rows read need to have been written — for rows below
 * min(t * inner, H - inner), t = number of complete tile rows, those are the first t * x_tiles patches (the last
 * `inner` rows belong to the clamped last tile row).  dev_patches / count describe the whole [count,C,P,P] buffer. */
int dsen2_recompose_rows(const float *dev_patches, int count, int C, int P, int border, float *dev_img, int H,
                         int W, float scale, int row0, int row1, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DSEN2_HIP_H */
