/*
 * TEST INFRASTRUCTURE ONLY — plain-C CPU restatement ("oracle") of the DSen2 hot path.
 * Not part of the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load the library built from this file (oracle/Makefile -> oracle/libdsen2_oracle.so).
 *
 * Restates, in double precision, the arithmetic of (paths relative to /root/reference):
 *   utils/DSen2Net.py:9-15    resBlock   x + 0.1 * conv(relu(conv(x)))
 *   utils/DSen2Net.py:18-43   s2model    concat -> conv+relu -> d x resBlock -> conv -> + last input
 *   utils/patches.py:11-16    interp_patches (skimage.transform.resize, order 1, mode='reflect')
 * with the keras Conv2D semantics the reference relies on: cross-correlation, 'same' = 1 px zero
 * pad, kernels stored HWIO (3,3,Cin,Cout), bias before activation.
 *
 * PARITY STATUS: unpinned for the CNN (keras/tensorflow absent, checkpoints stripped from the
 * reference checkout); the up-sampler is pinned against skimage 0.18.3 outputs captured in
 * tests/golden/ by tests/golden/make_golden_patches.py.
 *
 * All tensors are NCHW, C-order.  Weights arrive in "keras flat" order (see oracle/dsen2_oracle.py).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define RES_SCALE 0.1 /* utils/DSen2Net.py:9 */

/* out[n,o,y,x] = bias[o] + sum_{dy,dx,c} in[n,c,y+dy-1,x+dx-1] * k[dy,dx,c,o]   (zero outside) */
void dsen2_oracle_conv3x3_f64(const double *in, const float *kernel, const float *bias, double *out,
                              int n, int cin, int cout, int h, int w, int relu)
{
    const size_t plane = (size_t)h * w;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b) {
        for (int o = 0; o < cout; ++o) {
            double *dst = out + ((size_t)b * cout + o) * plane;
            for (size_t i = 0; i < plane; ++i) dst[i] = (double)bias[o];
            for (int c = 0; c < cin; ++c) {
                const double *src = in + ((size_t)b * cin + c) * plane;
                for (int dy = 0; dy < 3; ++dy) {
                    for (int dx = 0; dx < 3; ++dx) {
                        const double kv = (double)kernel[(((size_t)dy * 3 + dx) * cin + c) * cout + o];
                        const int y0 = dy == 0 ? 1 : 0, y1 = dy == 2 ? h - 1 : h;
                        const int x0 = dx == 0 ? 1 : 0, x1 = dx == 2 ? w - 1 : w;
                        for (int y = y0; y < y1; ++y) {
                            const double *s = src + (size_t)(y + dy - 1) * w + (dx - 1);
                            double *d = dst + (size_t)y * w;
                            for (int x = x0; x < x1; ++x) d[x] += s[x] * kv;
                        }
                    }
                }
            }
            if (relu)
                for (size_t i = 0; i < plane; ++i) dst[i] = dst[i] > 0.0 ? dst[i] : 0.0;
        }
    }
}

/*
 * s2model forward.  x = concat of the 2 or 3 inputs along channels, already assembled by the
 * caller as `xcat` [n, cin, h, w]; `skip` = the last (lowest-resolution) input [n, cout, h, w].
 * Returns 0 on success, -1 on allocation failure.
 */
int dsen2_oracle_forward_f64(const double *xcat, const double *skip, const float *flat, double *out,
                             int n, int cin, int cout, int h, int w, int num_layers, int feat)
{
    const size_t act = (size_t)n * feat * h * w;
    double *a = (double *)malloc(act * sizeof(double));
    double *t = (double *)malloc(act * sizeof(double));
    double *u = (double *)malloc(act * sizeof(double));
    if (!a || !t || !u) { free(a); free(t); free(u); return -1; }
    const float *p = flat;
    /* DSen2Net.py:29 */
    dsen2_oracle_conv3x3_f64(xcat, p, p + (size_t)9 * cin * feat, a, n, cin, feat, h, w, 1);
    p += (size_t)9 * cin * feat + feat;
    const size_t body = (size_t)9 * feat * feat;
    for (int l = 0; l < num_layers; ++l) { /* DSen2Net.py:31-32 -> :9-15 */
        dsen2_oracle_conv3x3_f64(a, p, p + body, t, n, feat, feat, h, w, 1);
        p += body + feat;
        dsen2_oracle_conv3x3_f64(t, p, p + body, u, n, feat, feat, h, w, 0);
        p += body + feat;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < act; ++i) a[i] = a[i] + u[i] * RES_SCALE;
    }
    /* DSen2Net.py:35,38,41 */
    dsen2_oracle_conv3x3_f64(a, p, p + (size_t)9 * feat * cout, out, n, feat, cout, h, w, 0);
    const size_t on = (size_t)n * cout * h * w;
    for (size_t i = 0; i < on; ++i) out[i] += skip[i];
    free(a); free(t); free(u);
    return 0;
}

/* skimage _warps_cy coord_map(mode='R'): mirror without repeating the edge sample. */
static int mirror_index(long i, int dim)
{
    const long cmax = dim - 1;
    if (dim == 1) return 0;
    if (i < 0) {
        const long k = -i;
        return (int)(((k / cmax) % 2 != 0) ? cmax - (k % cmax) : (k % cmax));
    }
    if (i > cmax) return (int)(((i / cmax) % 2 != 0) ? cmax - (i % cmax) : (i % cmax));
    return (int)i;
}

/*
 * interp_patches (utils/patches.py:11-16) in scikit-image 0.18.3's OWN arithmetic for a float32 image, operation by
 * operation (the second, independent restatement of oracle/patches_oracle.py's f32_coords mode; both reproduce the captured
 * outputs of the reference bit for bit, tests/test_oracle_cnn.py):
 *   warp() casts its matrix to float32:  c = float32(scale) * float32(j) + float32(offset), two rounded operations;
 *   bilinear_interpolation[float32]:     dc = c - floor(c) in float32;
 *       top = (1.0 - (double)dc) * (double)top_left + (double)(dc * top_right)     <- that product alone is float32 x float32
 *       out = (float)((1.0 - (double)dr) * top + (double)dr * bottom)
 *   then clipped to the plane's [min, max] (warp(clip=True); a no-op for this arithmetic, kept for fidelity) and * 30000.
 * `volatile` keeps every float32 intermediate a float32 whatever the compiler's evaluation method or contraction rules.
 */
void dsen2_oracle_upsample_skimage(const float *in, float *out, int planes, int h, int w, int oh, int ow)
{
    const float sy = (float)((double)h / oh), sx = (float)((double)w / ow);
    const float oy = (float)(0.5 * ((double)h / oh) - 0.5), ox = (float)(0.5 * ((double)w / ow) - 0.5);
#pragma omp parallel for schedule(static)
    for (int p = 0; p < planes; ++p) {
        const float *src = in + (size_t)p * h * w;
        float *dst = out + (size_t)p * oh * ow;
        float lo = src[0] / 30000.0f, hi = lo;
        for (size_t k = 1; k < (size_t)h * w; ++k) {
            const float q = src[k] / 30000.0f;
            lo = q < lo ? q : lo;
            hi = q > hi ? q : hi;
        }
        for (int i = 0; i < oh; ++i) {
            volatile float rm = sy * (float)i;
            volatile float r = rm + oy;
            const float rf = floorf(r);
            volatile float drf = r - rf;
            const double dr = (double)drf;
            const int ra = mirror_index((long)rf, h), rb = mirror_index((long)ceilf(r), h);
            for (int j = 0; j < ow; ++j) {
                volatile float cm = sx * (float)j;
                volatile float c = cm + ox;
                const float cf = floorf(c);
                volatile float dc = c - cf;
                const int ca = mirror_index((long)cf, w), cb = mirror_index((long)ceilf(c), w);
                volatile float tl = src[(size_t)ra * w + ca] / 30000.0f, tr = src[(size_t)ra * w + cb] / 30000.0f;
                volatile float bl = src[(size_t)rb * w + ca] / 30000.0f, br = src[(size_t)rb * w + cb] / 30000.0f;
                volatile float ptr_ = dc * tr, pbr = dc * br;                    /* float32 products */
                const double top = (1.0 - (double)dc) * (double)tl + (double)ptr_;
                const double bot = (1.0 - (double)dc) * (double)bl + (double)pbr;
                volatile float v = (float)((1.0 - dr) * top + dr * bot);
                v = v < lo ? lo : (v > hi ? hi : v);
                volatile float scaled = v * 30000.0f;
                dst[(size_t)i * ow + j] = scaled;
            }
        }
    }
}

/*
 * interp_patches (utils/patches.py:11-16): per plane, resize(x/30000, (oh,ow), mode='reflect')*30000.
 * Half-pixel-centre bilinear: src = (dst + 0.5) * (in/out) - 0.5; neighbours floor/ceil, mirrored.
 * Computed in double from the float32 inputs; result rounded to float32 like the reference's store.
 */
void dsen2_oracle_upsample_f64(const float *in, float *out, int planes, int h, int w, int oh, int ow)
{
    const double fy = (double)h / oh, fx = (double)w / ow;
#pragma omp parallel for schedule(static)
    for (int p = 0; p < planes; ++p) {
        const float *src = in + (size_t)p * h * w;
        float *dst = out + (size_t)p * oh * ow;
        for (int i = 0; i < oh; ++i) {
            const double r = (i + 0.5) * fy - 0.5;
            const long r0 = (long)floor(r), r1 = (long)ceil(r);
            const double dr = r - (double)r0;
            const int ra = mirror_index(r0, h), rb = mirror_index(r1, h);
            for (int j = 0; j < ow; ++j) {
                const double c = (j + 0.5) * fx - 0.5;
                const long c0 = (long)floor(c), c1 = (long)ceil(c);
                const double dc = c - (double)c0;
                const int ca = mirror_index(c0, w), cb = mirror_index(c1, w);
                const double tl = (double)(src[(size_t)ra * w + ca] / 30000.0f);
                const double tr = (double)(src[(size_t)ra * w + cb] / 30000.0f);
                const double bl = (double)(src[(size_t)rb * w + ca] / 30000.0f);
                const double br = (double)(src[(size_t)rb * w + cb] / 30000.0f);
                const double top = (1.0 - dc) * tl + dc * tr;
                const double bot = (1.0 - dc) * bl + dc * br;
                dst[(size_t)i * ow + j] = (float)((1.0 - dr) * top + dr * bot) * 30000.0f;
            }
        }
    }
}
