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
