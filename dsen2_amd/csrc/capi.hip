// capi.hip — the extern "C" boundary of libdsen2_hip.so (declared in include/dsen2_hip.h).
// Host-side orchestration only: argument checks, weight packing/upload, workspace carving and the
// launch sequence of one forward pass.  No torch types, no allocation inside the forward path.
#include "../../include/dsen2_hip.h"

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>
#include <vector>

#include "dsen2_internal.h"

using namespace dsen2;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// Nothing may leave an extern "C" entry point as a C++ exception (std::bad_alloc from a staging vector, std::system_error
// from a mutex): through a C / ctypes caller that is std::terminate -> abort() of the host process.  Every entry point
// that can allocate or lock runs its body through this and reports DSEN2_ERR_* with dsen2_last_error() instead.
template <class F>
int guarded(F&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return fail(DSEN2_ERR_NOMEM, "out of host memory");            // not the caller's arguments: its own code
  } catch (const std::exception& e) {
    return fail(DSEN2_ERR_INTERNAL, "unexpected C++ exception: %s", e.what());
  } catch (...) {
    return fail(DSEN2_ERR_INTERNAL, "unexpected C++ exception");
  }
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return fail(DSEN2_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

struct Layer {
  int cin, cout;        // real channel counts (keras)
  int epilogue;
  bool bf16;            // weights packed as bf16 for the bf16-operand body kernel (conv3x3_body16w.hip)
  bool x3;              // ... as the (wh, wl, wh) planes of the bf16x3 form (precision 2): 3 x the bf16 weights
  PackGeom geom;
  size_t w_off, b_off;  // float offsets inside dev_params
  size_t w16_off;       // first layer of a precision-1 / -2 model: its bf16 (wh | wl) form for conv3x3_first16.hip; 0 = none
  size_t flat_off;      // float offset of the kernel inside the keras-flat array
};

constexpr int kBf16ChunkChannels = 32;   // input channels per weight chunk of conv3x3_body16w.hip

#ifdef DSEN2_DIAG
// Diagnostic build only (tools/): defaults copied into models and single-layer calls made AFTER dsen2_diag_set.
// The product library has no mutable globals — a model's kernel structures are fixed constants.
Tuning g_diag_tuning;
unsigned long long* g_diag_stamps = nullptr;   // device buffer for ablation bit 32 (dsen2_diag_set_stamps)
#endif
Tuning default_tuning() {
#ifdef DSEN2_DIAG
  return g_diag_tuning;
#else
  return Tuning{};
#endif
}

constexpr int kWarmLaunches = 24;      // dsen2_model_time_body_conv: untimed launches before the timed ones
constexpr size_t kAlignFloats = 64;   // 256-byte alignment of every device sub-buffer
size_t align_up(size_t v) { return (v + kAlignFloats - 1) / kAlignFloats * kAlignFloats; }

}  // namespace

struct dsen2_model {
  int c10, c20, c60, cin, cout, num_layers, feat, precision;
  int device;
  Tuning tune;          // kernel structures, fixed at creation
  std::vector<Layer> layers;
  size_t n_params;
  size_t chain_stride;  // precision 1 / 2: bytes between the packed weights (= between the biases) of consecutive body layers; 0 = not uniform
  size_t dev_param_floats;
  float* dev_params;
  bool loaded;
};

// A handle belongs to the device that was current when it was created (its packed weights live there): a call made with
// another current device would hand device A's pointers to kernels launched on device B.
static int check_device(const dsen2_model* m) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return fail(DSEN2_ERR_NO_DEVICE, "no HIP device");
  if (dev != m->device)
    return fail(DSEN2_ERR_INVALID, "model handle belongs to device %d but the calling thread's current device is %d "
                "(one handle per device: hipSetDevice(%d) before the call)", m->device, dev, m->device);
  return DSEN2_OK;
}

extern "C" {

const char* dsen2_version(void) {
#ifdef DSEN2_DIAG
  return "dsen2_hip 0.3-diag (gfx950; fp32 MFMA 32x32x2 / bf16 MFMA 16x16x32 / bf16x3; DIAGNOSTIC build)";
#else
  return "dsen2_hip 0.3 (gfx950; fp32 MFMA 32x32x2 / bf16 MFMA 16x16x32 / bf16x3)";
#endif
}
const char* dsen2_last_error(void) { return g_err; }

#ifdef DSEN2_DIAG
// Diagnostic build only — not declared in include/dsen2_hip.h.  key 0: structure of the fp32 body convolution
// (14 default, 11-13 sub-variants of conv3x3_body32.hip, 0 one tile per workgroup); key 1: timing-only ablation
// mask of the persistent body kernels (outputs are WRONG while non-zero); key 2: output-layer kernel (2 = vector units,
// 0 = padded MFMA block).
int dsen2_diag_set(int key, int value) {
  if (key == 0) {
    if (value != 0 && (value < 11 || value > 14)) return fail(DSEN2_ERR_INVALID, "body variant %d unknown", value);
    g_diag_tuning.body_variant = value;
    return DSEN2_OK;
  }
  if (key == 1) {
    g_diag_tuning.ablate = value;
    return DSEN2_OK;
  }
  if (key == 2) {
    if (value != 0 && value != 2 && value != 3) return fail(DSEN2_ERR_INVALID, "output variant %d unknown", value);
    g_diag_tuning.out_variant = value;
    return DSEN2_OK;
  }
  if (key == 3) {   // at most `value` workgroups for the bf16 body kernel (0 = one per CU): per-CU vs chip-wide limits
    g_diag_tuning.grid_cap = value;
    return DSEN2_OK;
  }
  if (key == 5) {   // timing-only ablation mask of the first convolution (1 no stores, 2 no MFMAs, 4 no gather)
    g_diag_tuning.first_ablate = value;
    return DSEN2_OK;
  }
  if (key == 6) {   // timing-only ablation mask of the matrix-core output convolution (conv3x3_out_mfma.hip)
    g_diag_tuning.out_ablate = value;
    return DSEN2_OK;
  }
  if (key == 4) {   // 0 = always launch the bf16 body convolutions layer by layer (A/B against the chain kernel)
    g_diag_tuning.chain = value;
    return DSEN2_OK;
  }
  return fail(DSEN2_ERR_INVALID, "unknown diagnostic key %d", key);
}
// device buffer (>= 64 KiB) the stamping build of the bf16 body kernel (ablation mask 32) writes s_memtime values to
int dsen2_diag_set_stamps(void* dev_buffer) {
  g_diag_stamps = reinterpret_cast<unsigned long long*>(dev_buffer);
  return DSEN2_OK;
}
#endif

int dsen2_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(DSEN2_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  int good = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, i) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++good;
  }
  return good;
}

static int model_create_unguarded(dsen2_model** out, int c10, int c20, int c60, int num_layers, int feature_size,
                       int precision) {
  if (!out) return fail(DSEN2_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (c10 <= 0 || c20 <= 0 || c60 < 0 || num_layers < 0) return fail(DSEN2_ERR_INVALID, "bad channel/layer counts");
  if (feature_size != 128 && feature_size != 256)
    return fail(DSEN2_ERR_INVALID, "feature_size %d unsupported (128 or 256)", feature_size);
  if (precision < 0 || precision > 2)
    return fail(DSEN2_ERR_INVALID, "precision %d unknown (0 = fp32, 1 = bf16 operands, 2 = bf16x3)", precision);
  const int cin = c10 + c20 + c60;
  const int cout = c60 > 0 ? c60 : c20;   // utils/DSen2Net.py:35 — input_shape[-1][0]
  if (cin > 16) return fail(DSEN2_ERR_INVALID, "%d input channels > 16", cin);
  if (cout > 32) return fail(DSEN2_ERR_INVALID, "%d output channels > 32", cout);
  dsen2_model* m = new (std::nothrow) dsen2_model();
  if (!m) return fail(DSEN2_ERR_INVALID, "out of host memory");
  m->c10 = c10; m->c20 = c20; m->c60 = c60; m->cin = cin; m->cout = cout;
  m->num_layers = num_layers; m->feat = feature_size; m->precision = precision;
  m->dev_params = nullptr; m->loaded = false;
  m->tune = default_tuning();
  if (hipGetDevice(&m->device) != hipSuccess) {
    delete m;
    return fail(DSEN2_ERR_NO_DEVICE, "no HIP device");
  }
  // graph order of utils/DSen2Net.py:29-35
  std::vector<std::pair<int, int>> shapes;
  std::vector<int> epis;
  shapes.push_back({cin, feature_size}); epis.push_back(kEpiRelu);
  for (int i = 0; i < num_layers; ++i) {
    shapes.push_back({feature_size, feature_size}); epis.push_back(kEpiRelu);
    shapes.push_back({feature_size, feature_size}); epis.push_back(kEpiResidual);
  }
  shapes.push_back({feature_size, cout}); epis.push_back(kEpiSkipNCHW);
  size_t flat = 0, dev = 0;
  for (size_t i = 0; i < shapes.size(); ++i) {
    Layer L;
    L.cin = shapes[i].first; L.cout = shapes[i].second; L.epilogue = epis[i];
    if (!conv_pack_geometry(L.cin, L.cout, L.epilogue, m->tune, &L.geom)) {
      delete m;
      return fail(DSEN2_ERR_INVALID, "no kernel for conv %d->%d", L.cin, L.cout);
    }
    L.flat_off = flat;
    flat += (size_t)9 * L.cin * L.cout + L.cout;
    const bool body = L.cin == feature_size && L.cout == feature_size;             // residual-block convolutions only
    L.bf16 = precision == 1 && body;
    L.x3 = precision == 2 && body;
    L.w_off = dev;
    dev += align_up(L.bf16 ? (size_t)9 * L.cin * L.cout / 2 : L.x3 ? (size_t)27 * L.cin * L.cout / 2 : packed_weight_floats(L.geom));
    L.b_off = dev; dev += align_up((size_t)L.geom.cout_pad);
    // precision 1 / 2: the first convolution runs on the bf16 matrix cores (conv3x3_first16.hip) for the Sentinel-2 band
    // groups 4 + 6 (+ 2); its fp32 form above stays for the generic fallback
    L.w16_off = 0;
    if (i == 0 && precision != 0 && num_layers > 0 && c10 == 4 && c20 == 6 && (c60 == 0 || c60 == 2)) {
      L.w16_off = dev;
      dev += align_up((first16_weight_u16(L.cout, precision == 2) + 1) / 2);
    }
    m->layers.push_back(L);
  }
  m->n_params = flat;
  m->dev_param_floats = dev;
  // the chain kernel (one launch over all body layers) addresses layer l's weights and bias at l * chain_stride
  m->chain_stride = 0;
  if ((precision == 1 || precision == 2) && num_layers > 0) {
    const size_t stride = m->layers[2].w_off - m->layers[1].w_off;
    bool uniform = true;
    for (int l = 1; l <= 2 * num_layers; ++l)
      uniform = uniform && (m->layers[l].bf16 || m->layers[l].x3) && m->layers[l].w_off == m->layers[1].w_off + (size_t)(l - 1) * stride &&
                m->layers[l].b_off == m->layers[1].b_off + (size_t)(l - 1) * stride;
    if (uniform) m->chain_stride = stride * sizeof(float);
  }
  *out = m;
  return DSEN2_OK;
}

void dsen2_model_destroy(dsen2_model* m) {
  if (!m) return;
  if (m->dev_params) (void)hipFree(m->dev_params);
  delete m;
}

size_t dsen2_model_num_params(const dsen2_model* m) { return m ? m->n_params : 0; }

static int model_load_weights_unguarded(dsen2_model* m, const float* host_flat, size_t count) {
  if (!m || !host_flat) return fail(DSEN2_ERR_INVALID, "NULL argument");
  if (count != m->n_params)
    return fail(DSEN2_ERR_INVALID, "expected %zu parameters, got %zu", m->n_params, count);
  if (int rc = check_device(m)) return rc;      // the packed weights are allocated on the current device
  std::vector<float> staged(m->dev_param_floats, 0.f);
  for (const Layer& L : m->layers) {
    const float* k = host_flat + L.flat_off;
    const float* b = k + (size_t)9 * L.cin * L.cout;
    if (L.bf16)
      pack_conv_weights_bf16_host(k, L.cin, L.cout, kBf16ChunkChannels, true, reinterpret_cast<uint16_t*>(staged.data() + L.w_off));
    else if (L.x3)
      pack_conv_weights_bf16x3_host(k, L.cin, L.cout, reinterpret_cast<uint16_t*>(staged.data() + L.w_off));
    else
      pack_conv_weights_host(k, L.cin, L.cout, L.geom, staged.data() + L.w_off);
    memcpy(staged.data() + L.b_off, b, sizeof(float) * L.cout);
    if (L.w16_off) pack_first16_weights_host(k, L.cin, L.cout, m->precision == 2, reinterpret_cast<uint16_t*>(staged.data() + L.w16_off));
  }
  if (!m->dev_params) HIP_TRY(hipMalloc((void**)&m->dev_params, m->dev_param_floats * sizeof(float)));
  HIP_TRY(hipMemcpy(m->dev_params, staged.data(), m->dev_param_floats * sizeof(float), hipMemcpyHostToDevice));
  m->loaded = true;
  return DSEN2_OK;
}

int dsen2_model_workspace_bytes(const dsen2_model* m, int n, int h, int w, size_t* bytes) {
  if (!m || !bytes || n <= 0 || h <= 0 || w <= 0) return fail(DSEN2_ERR_INVALID, "bad argument");
  const size_t pix = (size_t)n * h * w;
  // fp32: x0 | a | t.   bf16: x0 | a (fp32: the last block's output) | hi | lo | t
  // (hi, lo: the residual stream as two 16-bit planes; t: bf16; each half an fp32 tensor)
  // bf16x3: x0 | a (fp32: the first convolution's and the last block's output) | hx (hi | xl planes) | lo16 | t (hi | lo planes)
  const size_t full = align_up(pix * m->feat), half = align_up(pix * m->feat / 2);
  *bytes = (align_up(pix * 16) + full + (m->precision == 1 ? 3 * half : m->precision == 2 ? 2 * full + half : full)) * sizeof(float);
  return DSEN2_OK;
}

static hipError_t launch_bf16_body(const ConvParams& p, int feat, int epilogue, const Tuning& t, hipStream_t stream) {
  // (masks from 1024 up belong to the chain kernel)
  return launch_conv3x3_body16w(p, feat, epilogue, t.ablate & 1023, stream, t.grid_cap);
}

static int check_shape(const dsen2_model* m, int n, int h, int w) {
  if (n <= 0 || h <= 0 || w <= 0) return fail(DSEN2_ERR_INVALID, "bad shape n=%d h=%d w=%d", n, h, w);
  if ((size_t)h * w * (size_t)(m ? m->feat : 256) >= ((size_t)1 << 29))
    return fail(DSEN2_ERR_INVALID, "one image of %dx%d exceeds 2^31 activation bytes", h, w);
  return DSEN2_OK;
}

static ConvParams make_params(const float* in, const float* wpk, const float* bias, const float* aux, float* out,
                              int n, int h, int w, int cout_real, float scale) {
  ConvParams p;
  p.in = in; p.wpk = wpk; p.bias = bias; p.aux = aux; p.out = out; p.out2 = nullptr;
  p.n = n; p.h = h; p.w = w;
  p.tiles_x = (w + kTile - 1) / kTile; p.tiles_y = (h + kTile - 1) / kTile;
  p.cout_real = cout_real; p.res_scale = scale; p.diag = nullptr;
  return p;
}

// ev (optional, 4 events): recorded on the stream before the first convolution, before the first and after the last
// residual-block convolution, and after the output convolution
static int forward_impl(dsen2_model* m, const float* x10, const float* x20, const float* x60, float* out, int n,
                        int h, int w, void* workspace, size_t workspace_bytes, void* stream_, const hipEvent_t* ev) {
  if (!m || !x10 || !x20 || !out || !workspace) return fail(DSEN2_ERR_INVALID, "NULL argument");
  const hipEvent_t ev_fwd0 = ev ? ev[0] : nullptr, ev_body0 = ev ? ev[1] : nullptr, ev_body1 = ev ? ev[2] : nullptr,
                   ev_fwd1 = ev ? ev[3] : nullptr;
  if ((m->c60 > 0) != (x60 != nullptr)) return fail(DSEN2_ERR_INVALID, "x60 must be given iff the model has a 60 m input");
  if (!m->loaded) return fail(DSEN2_ERR_NO_WEIGHTS, "dsen2_model_load_weights has not been called");
  int rc = check_shape(m, n, h, w);
  if (rc) return rc;
  rc = check_device(m);
  if (rc) return rc;
  size_t need = 0;
  dsen2_model_workspace_bytes(m, n, h, w, &need);
  if (workspace_bytes < need) return fail(DSEN2_ERR_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, need);
  hipStream_t stream = (hipStream_t)stream_;
  const size_t pix = (size_t)n * h * w;
  float* x0 = (float*)workspace;                 // NHWC16 packed input
  float* a = x0 + align_up(pix * 16);            // residual stream x
  float* t = a + align_up(pix * m->feat);        // relu(convA(x))
  const float* P = m->dev_params;
  const float* skip = m->c60 > 0 ? x60 : x20;    // utils/DSen2Net.py:38,41

  const int abl = m->tune.ablate;
  size_t li = 0;
  const bool planes = m->precision == 1 && m->num_layers > 0;
  const bool x3 = m->precision == 2 && m->num_layers > 0;
  const size_t ws_full = align_up(pix * m->feat), ws_half = align_up(pix * m->feat / 2);
  bool x3_stream_written = false;      // precision 2: the first convolution wrote the stream's tensors itself
  if (ev_fwd0) HIP_TRY(hipEventRecord(ev_fwd0, stream));
  {
    const Layer& L = m->layers[li++];            // DSen2Net.py:24-29: Concatenate + Conv2D + ReLU
    ConvParams pf = make_params(x0, P + L.w_off, P + L.b_off, nullptr, a, n, h, w, 0, 0.f);
    if (planes) {
      // a precision-1 model's first convolution writes the residual stream directly as its two blocked 16-bit planes
      pf.out = t;
      pf.out2 = t + ws_half;
    }
    const int epi0 = planes ? (int)kEpiReluSplit : L.epilogue;
    // the default structure reads the NCHW inputs itself (conv3x3_first.hip); other channel counts, and the reference
    // structure (variant 0), pack them to NHWC16 first
    hipError_t direct = hipErrorNotSupported;
    const FirstInputs fi{x60, m->c10, m->c20, m->c60};
    if (L.w16_off && (planes || x3)) {
      // precision 1 / 2: on the bf16 matrix cores, writing the residual stream's planes itself (conv3x3_first16.hip) —
      // precision 1: (hi, lo); precision 2: hx (hi | xl planes) and lo16
      ConvParams pd = pf;
      pd.in = x10;
      pd.aux = x20;
      pd.wpk = P + L.w16_off;
      pd.out = t;
      pd.out2 = x3 ? t + ws_full : t + ws_half;
      direct = launch_conv3x3_first16(pd, fi, m->feat, x3, stream);
      if (direct != hipSuccess && direct != hipErrorNotSupported)
        return fail(DSEN2_ERR_HIP, "first convolution (bf16 matrix cores) launch: %s", hipGetErrorString(direct));
      x3_stream_written = x3 && direct == hipSuccess;
    } else if (!planes && !x3 && (L.geom.variant == 10 || L.geom.variant == 12)) {
      ConvParams pd = pf;
      pd.in = x10;
      pd.aux = x20;
      direct = launch_conv3x3_first(pd, fi, m->feat, L.epilogue, stream, m->tune.first_ablate);
      if (direct != hipSuccess && direct != hipErrorNotSupported)
        return fail(DSEN2_ERR_HIP, "first convolution launch: %s", hipGetErrorString(direct));
    }
    if (direct != hipSuccess) {
      HIP_TRY(launch_pack_inputs(x10, x20, x60, m->c10, m->c20, m->c60, x0, n, h, w, stream));
      HIP_TRY(launch_conv3x3(pf, L.geom, epi0, 0, stream));
    }
  }
  if (m->precision == 2 && m->num_layers > 0) {
    // bf16x3 (conv3x3_body16w.hip, X3): fp32-grade products from three bf16 MFMAs.  The first convolution writes the stream's
    // tensors (hx = hi | xl planes, lo16) itself; conv-A reads hx, writes t (hi | lo planes); conv-B reads t, updates
    // (hx, lo16) in place — the last block's writes plain fp32 `a` for the (fp32) output convolution.
    void* hx = t;
    void* lo16 = t + ws_full;
    void* t2 = t + ws_full + ws_half;
    if (!x3_stream_written) HIP_TRY(launch_split3_f32(a, hx, lo16, n, h, w, m->feat, stream));   // (fallback first layer: fp32 `a`)
    if (ev_body0) HIP_TRY(hipEventRecord(ev_body0, stream));
    // one persistent launch over all 2d body convolutions when every CU gets whole patches (as for precision 1)
    hipError_t chained = hipErrorNotSupported;
    if (m->chain_stride != 0 && m->tune.chain && m->tune.grid_cap == 0 && m->tune.ablate == 0) {
      const Layer& L1 = m->layers[li];
      ConvParams pc = make_params(nullptr, P + L1.w_off, P + L1.b_off, nullptr, nullptr, n, h, w, 0, 0.1f);
      ChainArgs ca;
      ca.hi = hx; ca.lo = lo16; ca.t = t2; ca.out_f32 = a;
      ca.layer_stride = (unsigned)m->chain_stride; ca.n_layers = 2 * m->num_layers; ca.patches_per_wg = 0; ca.seamless = 0;
      chained = launch_conv3x3_body16w_chain(pc, ca, m->feat, stream, 0, true);
      if (chained == hipSuccess) li += 2 * (size_t)m->num_layers;
      else if (chained != hipErrorNotSupported) return fail(DSEN2_ERR_HIP, "bf16x3 chain kernel launch: %s", hipGetErrorString(chained));
    }
    for (int i = 0; i < m->num_layers && chained != hipSuccess; ++i) {
      const Layer& LA = m->layers[li++];
      HIP_TRY(launch_conv3x3_body16w_x3(make_params(reinterpret_cast<const float*>(hx), P + LA.w_off, P + LA.b_off, nullptr,
                                                    reinterpret_cast<float*>(t2), n, h, w, 0, 0.f), m->feat, kEpiRelu, stream));
      const Layer& LB = m->layers[li++];
      const bool last = i + 1 == m->num_layers;
      ConvParams pb = make_params(reinterpret_cast<const float*>(t2), P + LB.w_off, P + LB.b_off,
                                  reinterpret_cast<const float*>(hx), last ? a : reinterpret_cast<float*>(hx), n, h, w, 0, 0.1f);
      pb.out2 = lo16;
      HIP_TRY(launch_conv3x3_body16w_x3(pb, m->feat, last ? kEpiResidualF32 : kEpiResidual, stream));
    }
  } else if (planes) {
    // bf16 operands, fp32 accumulate, exact fp32 residual stream held as two 16-bit planes (hi = the bf16 operand of
    // the next convolution, lo = the low halves): conv-A reads hi, conv-B updates (hi, lo) in place; the last
    // block's conv-B writes plain fp32 for the (fp32) output convolution
    const size_t half = align_up(pix * m->feat / 2);
    void* hi = t;
    void* lo = t + half;
    void* tbf = t + 2 * half;
    if (ev_body0) HIP_TRY(hipEventRecord(ev_body0, stream));
    // One persistent launch over all 2d body convolutions when every CU gets whole patches (batch >= one patch per CU,
    // e.g. BASELINE configs[4]); hipErrorNotSupported = this batch keeps more CUs busy layer by layer.
    hipError_t chained = hipErrorNotSupported;
    if (m->chain_stride != 0 && m->tune.chain && m->tune.grid_cap == 0) {
      const Layer& L1 = m->layers[li];
      ConvParams pc = make_params(nullptr, P + L1.w_off, P + L1.b_off, nullptr, nullptr, n, h, w, 0, 0.1f);
#ifdef DSEN2_DIAG
      pc.diag = g_diag_stamps;
#endif
      ChainArgs ca;
      ca.hi = hi; ca.lo = lo; ca.t = tbf; ca.out_f32 = a;
      ca.layer_stride = (unsigned)m->chain_stride; ca.n_layers = 2 * m->num_layers; ca.patches_per_wg = 0; ca.seamless = 0;
      chained = launch_conv3x3_body16w_chain(pc, ca, m->feat, stream, m->tune.ablate);
      if (chained == hipSuccess) li += 2 * (size_t)m->num_layers;
      else if (chained != hipErrorNotSupported) return fail(DSEN2_ERR_HIP, "chain kernel launch: %s", hipGetErrorString(chained));
    }
    for (int i = 0; i < m->num_layers && chained != hipSuccess; ++i) {
      const Layer& LA = m->layers[li++];
      ConvParams pa = make_params(reinterpret_cast<const float*>(hi), P + LA.w_off, P + LA.b_off, nullptr,
                                  reinterpret_cast<float*>(tbf), n, h, w, 0, 0.f);
      HIP_TRY(launch_bf16_body(pa, m->feat, kEpiRelu, m->tune, stream));
      const Layer& LB = m->layers[li++];
      const bool last = i + 1 == m->num_layers;
      ConvParams pb = make_params(reinterpret_cast<const float*>(tbf), P + LB.w_off, P + LB.b_off,
                                  reinterpret_cast<const float*>(hi), last ? a : reinterpret_cast<float*>(hi), n, h, w, 0, 0.1f);
      pb.out2 = lo;
      HIP_TRY(launch_bf16_body(pb, m->feat, last ? kEpiResidualF32 : kEpiResidual, m->tune, stream));
    }
  } else {
    if (ev_body0) HIP_TRY(hipEventRecord(ev_body0, stream));
    for (int i = 0; i < m->num_layers; ++i) {      // DSen2Net.py:31-32 -> :9-15
      const Layer& LA = m->layers[li++];
      HIP_TRY(launch_conv3x3(make_params(a, P + LA.w_off, P + LA.b_off, nullptr, t, n, h, w, 0, 0.f), LA.geom, LA.epilogue, abl, stream));
      const Layer& LB = m->layers[li++];
      // in place on the residual stream: every workgroup reads aux and writes out at its own pixels only
      HIP_TRY(launch_conv3x3(make_params(t, P + LB.w_off, P + LB.b_off, a, a, n, h, w, 0, 0.1f), LB.geom, LB.epilogue, abl, stream));
    }
  }
  if (ev_body1) HIP_TRY(hipEventRecord(ev_body1, stream));
  {
    const Layer& L = m->layers[li++];            // DSen2Net.py:35,38,41
    ConvParams po = make_params(a, P + L.w_off, P + L.b_off, skip, out, n, h, w, m->cout, 0.f);
#ifdef DSEN2_DIAG
    po.diag = g_diag_stamps;
#endif
    HIP_TRY(launch_conv3x3(po, L.geom, L.epilogue, m->tune.out_ablate, stream));
  }
  if (ev_fwd1) HIP_TRY(hipEventRecord(ev_fwd1, stream));
  return DSEN2_OK;
}

static int model_body_launches_unguarded(const dsen2_model* m, int n, int h, int w) {
  if (!m) return fail(DSEN2_ERR_INVALID, "NULL model");
  int rc = check_shape(m, n, h, w);
  if (rc) return rc;
  rc = check_device(m);
  if (rc) return rc;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return fail(DSEN2_ERR_NO_DEVICE, "no HIP device");
  const bool chain = (m->precision == 1 || m->precision == 2) && m->num_layers > 0 && m->chain_stride != 0 && m->tune.chain && m->tune.grid_cap == 0 &&
                     body16w_chain_patches_per_wg(n, h, w, m->feat, cus) > 0;
  return chain ? 1 : 2 * m->num_layers;
}

static int model_forward_unguarded(dsen2_model* m, const float* x10, const float* x20, const float* x60, float* out, int n,
                        int h, int w, void* workspace, size_t workspace_bytes, void* stream) {
  return forward_impl(m, x10, x20, x60, out, n, h, w, workspace, workspace_bytes, stream, nullptr);
}

// `iters` forward passes with four events each (forward_impl); ms[0..3] = mean of (whole forward, first convolution,
// all residual-block convolutions, output convolution) — three consecutive intervals between the same four time stamps,
// so ms[1] + ms[2] + ms[3] = ms[0] by construction; ms[4] = host wall-clock per pass of this instrumented run (enqueue
// of the first pass to completion of the last, / iters): what the events themselves cost shows as ms[4] against an
// un-instrumented loop of the same passes.
//
// `warm` un-instrumented passes are enqueued right before the instrumented ones, with no synchronisation in between: after
// any idle stretch of a few milliseconds (a host-side allocation, a synchronisation followed by host work) this GPU needs
// ~25 launches of the body convolution (~25 ms) to come back to its steady clock — launches are up to 16 % slower meanwhile
// (profiles/r04_ablation.md §2: the per-dispatch timeline) — so intervals measured cold are not the running network's.
static int forward_profile_impl(dsen2_model* m, const float* x10, const float* x20, const float* x60, float* out, int n,
                                int h, int w, void* workspace, size_t workspace_bytes, void* stream, int warm, int iters,
                                float* ms) {
  if (iters <= 0 || iters > 4096 || warm < 0 || warm > 4096 || !ms) return fail(DSEN2_ERR_INVALID, "bad warm / iters / NULL result");
  if (!m) return fail(DSEN2_ERR_INVALID, "NULL model");
  // the events are destroyed on every path
  struct Events {
    std::vector<hipEvent_t> ev;
    ~Events() {
      for (hipEvent_t e : ev)
        if (e) (void)hipEventDestroy(e);
    }
  } evs;
  evs.ev.assign(4 * (size_t)iters, nullptr);
  for (hipEvent_t& e : evs.ev) HIP_TRY(hipEventCreate(&e));
  for (int i = 0; i < warm; ++i) {
    int rc = forward_impl(m, x10, x20, x60, out, n, h, w, workspace, workspace_bytes, stream, nullptr);
    if (rc != DSEN2_OK) return rc;
  }
  // (the host clock starts when the warm passes are enqueued, not when they finish: a synchronisation here would be the
  // idle stretch the warm passes exist to avoid; ms[4] is corrected for them below)
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; ++i) {
    int rc = forward_impl(m, x10, x20, x60, out, n, h, w, workspace, workspace_bytes, stream, &evs.ev[4 * (size_t)i]);
    if (rc != DSEN2_OK) {
      (void)hipStreamSynchronize((hipStream_t)stream);     // nothing recorded may outlive its event
      return rc;
    }
  }
  HIP_TRY(hipEventSynchronize(evs.ev.back()));
  const auto t1 = std::chrono::steady_clock::now();
  double sum[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = 0; i < iters; ++i) {
    const hipEvent_t* e = &evs.ev[4 * (size_t)i];
    float v = 0.f;
    HIP_TRY(hipEventElapsedTime(&v, e[0], e[3])); sum[0] += v;
    HIP_TRY(hipEventElapsedTime(&v, e[0], e[1])); sum[1] += v;
    HIP_TRY(hipEventElapsedTime(&v, e[1], e[2])); sum[2] += v;
    HIP_TRY(hipEventElapsedTime(&v, e[2], e[3])); sum[3] += v;
  }
  for (int k = 0; k < 4; ++k) ms[k] = (float)(sum[k] / iters);
  // host wall clock per instrumented pass: enqueue of the first instrumented pass to completion of the last one; when warm
  // passes are still running at t0, the interval from the first instrumented event (GPU clock) is the truthful one
  double wall = std::chrono::duration<double, std::milli>(t1 - t0).count();
  if (warm > 0) {
    float span = 0.f;
    HIP_TRY(hipEventElapsedTime(&span, evs.ev.front(), evs.ev.back()));
    wall = span;
  }
  ms[4] = (float)(wall / iters);
  return DSEN2_OK;
}

static int model_forward_timed_unguarded(dsen2_model* m, const float* x10, const float* x20, const float* x60, float* out, int n,
                              int h, int w, void* workspace, size_t workspace_bytes, void* stream, int iters,
                              float* body_ms_per_launch) {
  if (!body_ms_per_launch) return fail(DSEN2_ERR_INVALID, "NULL result");
  if (!m || m->num_layers <= 0) return fail(DSEN2_ERR_INVALID, "model has no residual blocks");
  float ms[5];
  int rc = forward_profile_impl(m, x10, x20, x60, out, n, h, w, workspace, workspace_bytes, stream, 0, iters, ms);
  if (rc == DSEN2_OK) *body_ms_per_launch = ms[2] / (2.0f * m->num_layers);
  return rc;
}

static int model_forward_profile_unguarded(dsen2_model* m, const float* x10, const float* x20, const float* x60, float* out,
                                           int n, int h, int w, void* workspace, size_t workspace_bytes, void* stream,
                                           int warm, int iters, float* ms5) {
  return forward_profile_impl(m, x10, x20, x60, out, n, h, w, workspace, workspace_bytes, stream, warm, iters, ms5);
}

static int conv3x3_nhwc_impl(const float* dev_in, const float* host_kernel, const float* host_bias, const float* dev_aux,
                             float* dev_out, int n, int h, int w, int cin, int cout, int epilogue, float res_scale,
                             void* stream_, const Tuning& tune) {
  if (!dev_in || !host_kernel || !host_bias || !dev_out) return fail(DSEN2_ERR_INVALID, "NULL argument");
  if (epilogue != kEpiRelu && epilogue != kEpiResidual && epilogue != kEpiSkipNCHW) return fail(DSEN2_ERR_INVALID, "epilogue %d", epilogue);
  if (epilogue != kEpiRelu && !dev_aux) return fail(DSEN2_ERR_INVALID, "epilogue %d needs dev_aux", epilogue);
  int rc = check_shape(nullptr, n, h, w);
  if (rc) return rc;
  PackGeom g;
  if (!conv_pack_geometry(cin, cout, epilogue, tune, &g) || g.cin_pad != cin)
    return fail(DSEN2_ERR_INVALID, "unsupported conv %d->%d epilogue %d", cin, cout, epilogue);
  hipStream_t stream = (hipStream_t)stream_;
  const size_t wf = packed_weight_floats(g);
  std::vector<float> staged(wf + g.cout_pad, 0.f);
  pack_conv_weights_host(host_kernel, cin, cout, g, staged.data());
  memcpy(staged.data() + wf, host_bias, sizeof(float) * cout);
  float* dev = nullptr;
  HIP_TRY(hipMalloc((void**)&dev, staged.size() * sizeof(float)));
  hipError_t e = hipMemcpy(dev, staged.data(), staged.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = launch_conv3x3(make_params(dev_in, dev, dev + wf, dev_aux, dev_out, n, h, w, cout, res_scale), g, epilogue, tune.ablate, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(dev);
  if (e != hipSuccess) return fail(DSEN2_ERR_HIP, "conv3x3 launch: %s", hipGetErrorString(e));
  return DSEN2_OK;
}

static int conv3x3_nhwc_unguarded(const float* dev_in, const float* host_kernel, const float* host_bias, const float* dev_aux,
                       float* dev_out, int n, int h, int w, int cin, int cout, int epilogue, float res_scale,
                       void* stream) {
  return conv3x3_nhwc_impl(dev_in, host_kernel, host_bias, dev_aux, dev_out, n, h, w, cin, cout, epilogue, res_scale,
                           stream, default_tuning());
}

static int conv3x3_nhwc_ref_unguarded(const float* dev_in, const float* host_kernel, const float* host_bias, const float* dev_aux,
                           float* dev_out, int n, int h, int w, int cin, int cout, int epilogue, float res_scale,
                           void* stream) {
  Tuning ref;                 // the one-tile-per-workgroup kernels of conv3x3_mfma.hip for every layer shape
  ref.body_variant = 0;
  ref.out_variant = 0;
  return conv3x3_nhwc_impl(dev_in, host_kernel, host_bias, dev_aux, dev_out, n, h, w, cin, cout, epilogue, res_scale,
                           stream, ref);
}

static int split_f32_unguarded(const float* dev_in, void* dev_hi, void* dev_lo, int n, int h, int w, int c, void* stream) {
  if (!dev_in || !dev_hi || !dev_lo || n < 0 || h <= 0 || w <= 0 || c <= 0 || c % 8 != 0 || c > 512)
    return fail(DSEN2_ERR_INVALID, "bad argument (c must be a multiple of 8, at most 512)");
  if (n == 0) return DSEN2_OK;
  HIP_TRY(launch_split_f32(dev_in, dev_hi, dev_lo, n, h, w, c, (hipStream_t)stream));
  return DSEN2_OK;
}

static int join_f32_unguarded(const void* dev_hi, const void* dev_lo, float* dev_out, int n, int h, int w, int c, void* stream) {
  if (!dev_out || !dev_hi || !dev_lo || n < 0 || h <= 0 || w <= 0 || c <= 0 || c % 8 != 0 || c > 512)
    return fail(DSEN2_ERR_INVALID, "bad argument (c must be a multiple of 8, at most 512)");
  if (n == 0) return DSEN2_OK;
  HIP_TRY(launch_join_f32(dev_hi, dev_lo, dev_out, n, h, w, c, (hipStream_t)stream));
  return DSEN2_OK;
}

static int split3_f32_unguarded(const float* dev_in, void* dev_hx, void* dev_lo, int n, int h, int w, int c, void* stream) {
  if (!dev_in || !dev_hx || !dev_lo || n < 0 || h <= 0 || w <= 0 || c <= 0 || c % 8 != 0 || c > 512)
    return fail(DSEN2_ERR_INVALID, "bad argument (c must be a multiple of 8, at most 512)");
  if (n == 0) return DSEN2_OK;
  HIP_TRY(launch_split3_f32(dev_in, dev_hx, dev_lo, n, h, w, c, (hipStream_t)stream));
  return DSEN2_OK;
}

static int conv3x3_body_bf16x3_unguarded(const void* dev_in_planes, const float* host_kernel, const float* host_bias,
                                         void* dev_res_hx, void* dev_res_lo, void* dev_out, int n, int h, int w, int feat,
                                         int epilogue, float res_scale, void* stream_) {
  if (!dev_in_planes || !host_kernel || !host_bias) return fail(DSEN2_ERR_INVALID, "NULL argument");
  if (feat != 128 && feat != 256) return fail(DSEN2_ERR_INVALID, "feat %d unsupported", feat);
  if (epilogue != kEpiRelu && epilogue != kEpiResidual && epilogue != kEpiResidualF32) return fail(DSEN2_ERR_INVALID, "epilogue %d", epilogue);
  if (epilogue != kEpiRelu && (!dev_res_hx || !dev_res_lo)) return fail(DSEN2_ERR_INVALID, "residual epilogue needs the stream's tensors");
  if (epilogue != kEpiResidual && !dev_out) return fail(DSEN2_ERR_INVALID, "dev_out is NULL");
  int rc = check_shape(nullptr, n, h, w);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const size_t wn = (size_t)27 * feat * feat;
  std::vector<uint16_t> wb(wn);
  pack_conv_weights_bf16x3_host(host_kernel, feat, feat, wb.data());
  char* dev = nullptr;
  HIP_TRY(hipMalloc((void**)&dev, wn * 2 + feat * sizeof(float)));
  hipError_t e = hipMemcpy(dev, wb.data(), wn * 2, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dev + wn * 2, host_bias, feat * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    ConvParams p = make_params(reinterpret_cast<const float*>(dev_in_planes), reinterpret_cast<const float*>(dev),
                               reinterpret_cast<const float*>(dev + wn * 2), reinterpret_cast<const float*>(dev_res_hx),
                               reinterpret_cast<float*>(epilogue == kEpiResidual ? dev_res_hx : dev_out), n, h, w, 0, res_scale);
    p.out2 = dev_res_lo;
    e = launch_conv3x3_body16w_x3(p, feat, epilogue, stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(dev);
  if (e != hipSuccess) return fail(DSEN2_ERR_HIP, "bf16x3 conv launch: %s", hipGetErrorString(e));
  return DSEN2_OK;
}

static int conv3x3_first_planes_unguarded(const float* dev_x10, const float* dev_x20, const float* dev_x60, int c10, int c20, int c60,
                                          const float* host_kernel, const float* host_bias, int feat, int precision,
                                          void* dev_out, void* dev_out2, int n, int h, int w, void* stream_) {
  if (!dev_x10 || !dev_x20 || !host_kernel || !host_bias || !dev_out || !dev_out2) return fail(DSEN2_ERR_INVALID, "NULL argument");
  if ((c60 > 0) != (dev_x60 != nullptr)) return fail(DSEN2_ERR_INVALID, "dev_x60 must be given iff c60 > 0");
  if (feat != 128 && feat != 256) return fail(DSEN2_ERR_INVALID, "feat %d unsupported", feat);
  if (precision != 1 && precision != 2) return fail(DSEN2_ERR_INVALID, "precision %d (1 = bf16 operands, 2 = bf16x3)", precision);
  if (c10 != 4 || c20 != 6 || (c60 != 0 && c60 != 2)) return fail(DSEN2_ERR_INVALID, "band groups %d + %d + %d (4 + 6 (+ 2) only)", c10, c20, c60);
  int rc = check_shape(nullptr, n, h, w);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const bool x3 = precision == 2;
  const int cin = c10 + c20 + c60;
  const size_t wn = first16_weight_u16(feat, x3);
  std::vector<uint16_t> wb(wn);
  pack_first16_weights_host(host_kernel, cin, feat, x3, wb.data());
  char* dev = nullptr;
  HIP_TRY(hipMalloc((void**)&dev, wn * 2 + feat * sizeof(float)));
  hipError_t e = hipMemcpy(dev, wb.data(), wn * 2, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dev + wn * 2, host_bias, feat * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    ConvParams p = make_params(dev_x10, reinterpret_cast<const float*>(dev), reinterpret_cast<const float*>(dev + wn * 2), dev_x20,
                               reinterpret_cast<float*>(dev_out), n, h, w, 0, 0.f);
    p.out2 = dev_out2;
    e = launch_conv3x3_first16(p, FirstInputs{dev_x60, c10, c20, c60}, feat, x3, stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(dev);
  if (e != hipSuccess) return fail(DSEN2_ERR_HIP, "first convolution (bf16 matrix cores) launch: %s", hipGetErrorString(e));
  return DSEN2_OK;
}

static int conv3x3_body_bf16_unguarded(const void* dev_in_bf16, const float* host_kernel, const float* host_bias, void* dev_res_hi,
                            void* dev_res_lo, void* dev_out, int n, int h, int w, int feat, int epilogue,
                            float res_scale, void* stream_) {
  if (!dev_in_bf16 || !host_kernel || !host_bias) return fail(DSEN2_ERR_INVALID, "NULL argument");
  if (feat != 128 && feat != 256) return fail(DSEN2_ERR_INVALID, "feat %d unsupported", feat);
  if (epilogue != kEpiRelu && epilogue != kEpiResidual && epilogue != kEpiResidualF32) return fail(DSEN2_ERR_INVALID, "epilogue %d", epilogue);
  if (epilogue != kEpiRelu && (!dev_res_hi || !dev_res_lo)) return fail(DSEN2_ERR_INVALID, "residual epilogue needs the hi and lo planes");
  if (epilogue != kEpiResidual && !dev_out) return fail(DSEN2_ERR_INVALID, "dev_out is NULL");
  int rc = check_shape(nullptr, n, h, w);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const size_t wn = (size_t)9 * feat * feat;
  std::vector<uint16_t> wb(wn);
  pack_conv_weights_bf16_host(host_kernel, feat, feat, kBf16ChunkChannels, true, wb.data());
  char* dev = nullptr;
  HIP_TRY(hipMalloc((void**)&dev, wn * 2 + feat * sizeof(float)));
  hipError_t e = hipMemcpy(dev, wb.data(), wn * 2, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dev + wn * 2, host_bias, feat * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    ConvParams p = make_params(reinterpret_cast<const float*>(dev_in_bf16), reinterpret_cast<const float*>(dev),
                               reinterpret_cast<const float*>(dev + wn * 2), reinterpret_cast<const float*>(dev_res_hi),
                               reinterpret_cast<float*>(epilogue == kEpiResidual ? dev_res_hi : dev_out), n, h, w, 0, res_scale);
    p.out2 = dev_res_lo;
    e = launch_bf16_body(p, feat, epilogue, default_tuning(), stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(dev);
  if (e != hipSuccess) return fail(DSEN2_ERR_HIP, "bf16 conv launch: %s", hipGetErrorString(e));
  return DSEN2_OK;
}

static int model_time_body_conv_unguarded(dsen2_model* m, int layer, const float* dev_in, const float* dev_aux, float* dev_out,
                               int n, int h, int w, int iters, void* stream_, float* ms_per_launch) {
  if (!m || !dev_in || !dev_out || !ms_per_launch || iters <= 0) return fail(DSEN2_ERR_INVALID, "bad argument");
  if (!m->loaded) return fail(DSEN2_ERR_NO_WEIGHTS, "weights not loaded");
  if (layer < 1 || layer > 2 * m->num_layers) return fail(DSEN2_ERR_INVALID, "layer %d out of range", layer);
  int rc = check_shape(m, n, h, w);
  if (rc) return rc;
  rc = check_device(m);
  if (rc) return rc;
  const Layer& L = m->layers[layer];
  if (L.x3) return fail(DSEN2_ERR_INVALID, "dsen2_model_time_body_conv: not available for precision 2 (use dsen2_model_forward_profile)");
  if (L.epilogue == kEpiResidual && !dev_aux) return fail(DSEN2_ERR_INVALID, "residual layer needs dev_aux");
  hipStream_t stream = (hipStream_t)stream_;
  const float* P = m->dev_params;
  ConvParams p = make_params(dev_in, P + L.w_off, P + L.b_off, dev_aux, dev_out, n, h, w, 0, 0.1f);
  int epi = L.epilogue;
  if (L.bf16 && L.epilogue == kEpiResidual) {
    // dev_aux = hi plane followed by lo plane (one fp32-sized buffer), updated in place; the last block's layer
    // writes fp32 to dev_out instead
    const bool last = layer == 2 * m->num_layers;
    p.out2 = reinterpret_cast<char*>(const_cast<float*>(dev_aux)) + (size_t)n * h * w * m->feat * 2;
    p.out = last ? dev_out : const_cast<float*>(dev_aux);
    epi = last ? kEpiResidualF32 : kEpiResidual;
  }
  const int abl = m->tune.ablate;
#ifdef DSEN2_DIAG
  p.diag = g_diag_stamps;
#endif
  auto launch = [&]() -> hipError_t {
    return L.bf16 ? launch_bf16_body(p, m->feat, epi, m->tune, stream) : launch_conv3x3(p, L.geom, L.epilogue, abl, stream);
  };
  // the two events are destroyed on every path
  struct Events {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Events() {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
    }
  } ev;
  HIP_TRY(hipEventCreate(&ev.e0));
  HIP_TRY(hipEventCreate(&ev.e1));
  // warm-up: ~25 ms of this kernel bring the chip back to its steady clock after an idle stretch (see forward_profile_impl)
  for (int i = 0; i < kWarmLaunches; ++i) HIP_TRY(launch());
  HIP_TRY(hipEventRecord(ev.e0, stream));
  for (int i = 0; i < iters; ++i) HIP_TRY(launch());
  HIP_TRY(hipEventRecord(ev.e1, stream));
  HIP_TRY(hipEventSynchronize(ev.e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, ev.e0, ev.e1));
  *ms_per_launch = ms / iters;
  return DSEN2_OK;
}

static int upsample_impl(const float* dev_in, float* dev_out, int planes, int h, int w, int oh, int ow, float post_divisor,
                         void* stream, bool general) {
  if (!dev_in || !dev_out || planes < 0 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || post_divisor == 0.f)
    return fail(DSEN2_ERR_INVALID, "bad argument");
  if ((size_t)h * w >= ((size_t)1 << 31) || (size_t)oh * ow >= ((size_t)1 << 31))
    return fail(DSEN2_ERR_INVALID, "plane too large");
  if (planes == 0) return DSEN2_OK;
  HIP_TRY(launch_upsample(dev_in, dev_out, planes, h, w, oh, ow, post_divisor, (hipStream_t)stream, general));
  return DSEN2_OK;
}

static int upsample_mirror_bilinear_unguarded(const float* dev_in, float* dev_out, int planes, int h, int w, int oh, int ow,
                                   float post_divisor, void* stream) {
  return upsample_impl(dev_in, dev_out, planes, h, w, oh, ow, post_divisor, stream, false);
}

static int upsample_mirror_bilinear_ref_unguarded(const float* dev_in, float* dev_out, int planes, int h, int w, int oh, int ow,
                                       float post_divisor, void* stream) {
  return upsample_impl(dev_in, dev_out, planes, h, w, oh, ow, post_divisor, stream, true);
}

static int tile_gather_unguarded(const float* dev_img, int H, int W, int C, int border, const int* dev_origins, int count, int P,
                      float divisor, float* dev_patches, void* stream) {
  if (!dev_img || !dev_patches || (!dev_origins && count > 0) || H <= 0 || W <= 0 || C <= 0 || border < 0 ||
      count < 0 || P <= 0 || divisor == 0.f)
    return fail(DSEN2_ERR_INVALID, "bad argument");
  if (border > H || border > W) return fail(DSEN2_ERR_INVALID, "border %d larger than the image", border);
  if ((size_t)C * P * P >= ((size_t)1 << 31)) return fail(DSEN2_ERR_INVALID, "patch too large");
  HIP_TRY(launch_tile_gather(dev_img, H, W, C, border, dev_origins, count, P, divisor, dev_patches, (hipStream_t)stream));
  return DSEN2_OK;
}

static int recompose_rows_unguarded(const float* dev_patches, int count, int C, int P, int border, float* dev_img, int H, int W,
                                    float scale, int row0, int row1, void* stream) {
  if (!dev_patches || !dev_img || count <= 0 || C <= 0 || P <= 0 || border < 0 || H <= 0 || W <= 0)
    return fail(DSEN2_ERR_INVALID, "bad argument");
  hipError_t e = launch_recompose(dev_patches, count, C, P, border, dev_img, H, W, scale, row0, row1, (hipStream_t)stream);
  if (e == hipErrorInvalidValue)
    return fail(DSEN2_ERR_INVALID, "recompose geometry: count=%d P=%d border=%d H=%d W=%d rows [%d, %d)", count, P, border, H, W, row0, row1);
  if (e != hipSuccess) return fail(DSEN2_ERR_HIP, "recompose launch: %s", hipGetErrorString(e));
  return DSEN2_OK;
}

static int recompose_unguarded(const float* dev_patches, int count, int C, int P, int border, float* dev_img, int H, int W,
                    float scale, void* stream) {
  if (!dev_patches || !dev_img || count <= 0 || C <= 0 || P <= 0 || border < 0 || H <= 0 || W <= 0)
    return fail(DSEN2_ERR_INVALID, "bad argument");
  hipError_t e = launch_recompose(dev_patches, count, C, P, border, dev_img, H, W, scale, 0, H, (hipStream_t)stream);
  if (e == hipErrorInvalidValue)
    return fail(DSEN2_ERR_INVALID, "recompose geometry: count=%d P=%d border=%d H=%d W=%d", count, P, border, H, W);
  if (e != hipSuccess) return fail(DSEN2_ERR_HIP, "recompose launch: %s", hipGetErrorString(e));
  return DSEN2_OK;
}

// ---- the exported forms of the entry points above: no C++ exception crosses the ABI (guarded(), top of this file) ----
int dsen2_model_create(dsen2_model** out, int c10, int c20, int c60, int num_layers, int feature_size, int precision) {
  return guarded([&] { return model_create_unguarded(out, c10, c20, c60, num_layers, feature_size, precision); });
}
int dsen2_model_load_weights(dsen2_model* m, const float* host_flat, size_t count) {
  return guarded([&] { return model_load_weights_unguarded(m, host_flat, count); });
}
int dsen2_model_forward(dsen2_model* m, const float* x10, const float* x20, const float* x60, float* out, int n, int h, int w, void* workspace, size_t workspace_bytes, void* stream) {
  return guarded([&] { return model_forward_unguarded(m, x10, x20, x60, out, n, h, w, workspace, workspace_bytes, stream); });
}
int dsen2_model_forward_timed(dsen2_model* m, const float* x10, const float* x20, const float* x60, float* out, int n, int h, int w, void* workspace, size_t workspace_bytes, void* stream, int iters, float* body_ms_per_launch) {
  return guarded([&] { return model_forward_timed_unguarded(m, x10, x20, x60, out, n, h, w, workspace, workspace_bytes, stream, iters, body_ms_per_launch); });
}
int dsen2_model_forward_profile(dsen2_model* m, const float* x10, const float* x20, const float* x60, float* out, int n, int h, int w, void* workspace, size_t workspace_bytes, void* stream, int warm, int iters, float* ms5) {
  return guarded([&] { return model_forward_profile_unguarded(m, x10, x20, x60, out, n, h, w, workspace, workspace_bytes, stream, warm, iters, ms5); });
}
int dsen2_conv3x3_nhwc(const float* dev_in, const float* host_kernel, const float* host_bias, const float* dev_aux, float* dev_out, int n, int h, int w, int cin, int cout, int epilogue, float res_scale, void* stream) {
  return guarded([&] { return conv3x3_nhwc_unguarded(dev_in, host_kernel, host_bias, dev_aux, dev_out, n, h, w, cin, cout, epilogue, res_scale, stream); });
}
int dsen2_conv3x3_nhwc_ref(const float* dev_in, const float* host_kernel, const float* host_bias, const float* dev_aux, float* dev_out, int n, int h, int w, int cin, int cout, int epilogue, float res_scale, void* stream) {
  return guarded([&] { return conv3x3_nhwc_ref_unguarded(dev_in, host_kernel, host_bias, dev_aux, dev_out, n, h, w, cin, cout, epilogue, res_scale, stream); });
}
int dsen2_split_f32(const float* dev_in, void* dev_hi, void* dev_lo, int n, int h, int w, int c, void* stream) {
  return guarded([&] { return split_f32_unguarded(dev_in, dev_hi, dev_lo, n, h, w, c, stream); });
}
int dsen2_join_f32(const void* dev_hi, const void* dev_lo, float* dev_out, int n, int h, int w, int c, void* stream) {
  return guarded([&] { return join_f32_unguarded(dev_hi, dev_lo, dev_out, n, h, w, c, stream); });
}
int dsen2_split3_f32(const float* dev_in, void* dev_hx, void* dev_lo, int n, int h, int w, int c, void* stream) {
  return guarded([&] { return split3_f32_unguarded(dev_in, dev_hx, dev_lo, n, h, w, c, stream); });
}
int dsen2_conv3x3_body_bf16x3(const void* dev_in_planes, const float* host_kernel, const float* host_bias, void* dev_res_hx, void* dev_res_lo, void* dev_out, int n, int h, int w, int feat, int epilogue, float res_scale, void* stream) {
  return guarded([&] { return conv3x3_body_bf16x3_unguarded(dev_in_planes, host_kernel, host_bias, dev_res_hx, dev_res_lo, dev_out, n, h, w, feat, epilogue, res_scale, stream); });
}
int dsen2_conv3x3_first_planes(const float* dev_x10, const float* dev_x20, const float* dev_x60, int c10, int c20, int c60, const float* host_kernel, const float* host_bias, int feat, int precision, void* dev_out, void* dev_out2, int n, int h, int w, void* stream) {
  return guarded([&] { return conv3x3_first_planes_unguarded(dev_x10, dev_x20, dev_x60, c10, c20, c60, host_kernel, host_bias, feat, precision, dev_out, dev_out2, n, h, w, stream); });
}
int dsen2_conv3x3_body_bf16(const void* dev_in_bf16, const float* host_kernel, const float* host_bias, void* dev_res_hi, void* dev_res_lo, void* dev_out, int n, int h, int w, int feat, int epilogue, float res_scale, void* stream_) {
  return guarded([&] { return conv3x3_body_bf16_unguarded(dev_in_bf16, host_kernel, host_bias, dev_res_hi, dev_res_lo, dev_out, n, h, w, feat, epilogue, res_scale, stream_); });
}
int dsen2_model_time_body_conv(dsen2_model* m, int layer, const float* dev_in, const float* dev_aux, float* dev_out, int n, int h, int w, int iters, void* stream_, float* ms_per_launch) {
  return guarded([&] { return model_time_body_conv_unguarded(m, layer, dev_in, dev_aux, dev_out, n, h, w, iters, stream_, ms_per_launch); });
}
int dsen2_upsample_mirror_bilinear(const float* dev_in, float* dev_out, int planes, int h, int w, int oh, int ow, float post_divisor, void* stream) {
  return guarded([&] { return upsample_mirror_bilinear_unguarded(dev_in, dev_out, planes, h, w, oh, ow, post_divisor, stream); });
}
int dsen2_upsample_mirror_bilinear_ref(const float* dev_in, float* dev_out, int planes, int h, int w, int oh, int ow, float post_divisor, void* stream) {
  return guarded([&] { return upsample_mirror_bilinear_ref_unguarded(dev_in, dev_out, planes, h, w, oh, ow, post_divisor, stream); });
}
int dsen2_tile_gather(const float* dev_img, int H, int W, int C, int border, const int* dev_origins, int count, int P, float divisor, float* dev_patches, void* stream) {
  return guarded([&] { return tile_gather_unguarded(dev_img, H, W, C, border, dev_origins, count, P, divisor, dev_patches, stream); });
}
int dsen2_recompose(const float* dev_patches, int count, int C, int P, int border, float* dev_img, int H, int W, float scale, void* stream) {
  return guarded([&] { return recompose_unguarded(dev_patches, count, C, P, border, dev_img, H, W, scale, stream); });
}
int dsen2_recompose_rows(const float* dev_patches, int count, int C, int P, int border, float* dev_img, int H, int W, float scale, int row0, int row1, void* stream) {
  return guarded([&] { return recompose_rows_unguarded(dev_patches, count, C, P, border, dev_img, H, W, scale, row0, row1, stream); });
}
int dsen2_model_body_launches(const dsen2_model* m, int n, int h, int w) {
  return guarded([&] { return model_body_launches_unguarded(m, n, h, w); });
}

}  // extern "C"
