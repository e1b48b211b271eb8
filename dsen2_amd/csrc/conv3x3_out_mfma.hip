// conv3x3_out_mfma.hip — the network's last convolution on the MATRIX cores: F -> Cout (6 or 2 channels) 3x3 'same' + bias
// + the low-resolution skip input, NHWC in, NCHW out (utils/DSen2Net.py:35,38,41).
//
// A 3x3 convolution with 6 outputs is a poor MFMA shape tap by tap (a 32-wide block is 81 % padding; conv3x3_out.hip therefore
// runs on the vector units, 122 us at the bench config for 7.25 GFLOP).  Here the nine taps are moved into the GEMM's M side:
//     P[(dy, dx, co)][pixel] = sum_c K[dy][dx][c][co] * X[pixel][c]          (9 * Cout = 54 of 64 rows used, 18 of 32 for Cout = 2)
//     out[y][x][co]          = sum_{dy, dx} P[(dy, dx, co)][(y + dy - 1, x + dx - 1)]
// i.e. ONE 1x1 convolution to 54 channels per input pixel on v_mfma_f32_32x32x2_f32, then nine shifted adds:
//   * a wave takes image rows; a row is cut into blocks of 32 consecutive pixels = the MFMA's N side (lane = pixel).  The B
//     operand (lane = pixel, k = lane half) comes straight from global memory: per unit (= 64 channels, 64 MFMAs) lane
//     (pixel, half) reads 32 consecutive channels of its pixel as 8 buffer_load_dwordx4, one unit ahead of the MFMAs; pixels
//     beyond the row's end read zeros through the descriptor's bound.  Weights: LDS, ds_read_b128 two steps ahead, waits counted by hand.
//   * the dx sum stays in registers: a lane holds P for its pixel and half of the output channels (rows are packed so that
//     lane half h owns co = h*CL .. h*CL + CL - 1), the neighbours' values arrive by DPP wave shifts; across a block edge the
//     value comes from the previous / next block of the same wave (ds_bpermute of the edge lane), which is why a block is
//     finished one block late.  Q[dy][co][x] = (P(dx=0)[x-1] + P(dx=1)[x]) + P(dx=2)[x+1] goes to an LDS ring of rows.
//   * the dy sum is a second stage after a workgroup barrier: out = (((Q[y-1][0] + Q[y][1]) + Q[y+1][2]) + bias) + skip, the
//     skip values fetched before the row's MFMAs.  The order of every sum is fixed by (y, x) alone — results do not depend on
//     how the image was cut into strips or blocks.
// A workgroup takes whole images (or strips of rows with one recomputed row above and below when there are few images).
// Bench config (512 x 32 x 32, F = 128, Cout 6): 79 us against 121 on the vector units; FETCH_SIZE = the algorithmic 281 MB.
// What bounds it and what was tried: profiles/archive/r03_ablation.md §4, HISTORY.md §3.2.
#include "dsen2_internal.h"

namespace dsen2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace outm {
constexpr int THREADS = 512;
constexpr int NPRE = 12;                       // skip values a thread fetches ahead of a phase
constexpr size_t LDS_LIMIT = 160 * 1024;
}  // namespace outm

struct OutMfmaGeom {
  int nb;           // 32-pixel blocks per row
  int wpad;         // 32 * nb
  int rpw;          // consecutive rows per wave and phase
  int ring;         // rows of Q in LDS: 8 * rpw + 2
  int strip_rows;   // output rows per job
  int strips;       // jobs per image
  int njobs;
  int ablate;       // DSEN2_DIAG builds: timing-only mask (1 no second stage, 2 no fetches, 4 no MFMAs, 8 no dx sum, 16 no barriers)
};

template <int F, int CL>
__global__ __launch_bounds__(outm::THREADS, 1) void conv3x3_out_mfma_kernel(const ConvParams p, const OutMfmaGeom g) {
  using namespace outm;
#ifdef DSEN2_DIAG
  const int abl = g.ablate;
#else
  constexpr int abl = 0;
#endif
  auto stamp = [&](int i) __attribute__((always_inline)) {      // mask 32: s_memtime of every wave of workgroups 0-3 (tools/stamp_out_conv.py)
    if ((abl & 32) && p.diag && blockIdx.x < 4 && (threadIdx.x & 63) == 0 && i < 32)
      p.diag[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 32 + i] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);
  constexpr int NRB = CL == 3 ? 2 : 1;         // 32-row blocks of P
  constexpr int NU = F / 64;                   // units of 64 channels (32 per lane half)
  constexpr int NJ = 8;                        // dwordx4 per lane and unit = groups of 4 MFMA steps
  constexpr int CO = 2 * CL;                   // Q planes per dy
  constexpr int NH = 3 * CL;                   // (dy, c) values a lane carries
  constexpr int W_FLOATS = NRB * NU * NJ * 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const w_s = smem;                     // [blk][u][jj][half][m] x 4 channels
  float* const b_s = smem + W_FLOATS;          // bias, 8 floats
  float* const q_s = b_s + 8;                  // [ring][dy][co][wpad]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pix = lane & 31, half = lane >> 5;
  const bool edge_l = pix == 0, edge_r = pix == 31;
  const int first_addr = (half * 32) * 4, last_addr = (half * 32 + 31) * 4;     // ds_bpermute source lanes (bytes)

  const int phase_rows = 8 * g.rpw;
  struct Job { int img, ys, ye, rlo, rhi; };
  auto job_of = [&](int job) -> Job {
    const int img = job / g.strips, s = job - img * g.strips;
    const int ys = s * g.strip_rows, ye = min(ys + g.strip_rows, p.h);
    return Job{img, ys, ye, max(ys - 1, 0), min(ye + 1, p.h)};
  };

  // ---- the wave's stream of units (job, phase, row, block, unit), one unit fetched ahead ----
  struct Cursor { int job, img, rhi, rb, k, j, u; bool valid; };
  Cursor nx;
  auto settle = [&]() {          // (rb, k = 0): move on to the next phase / job in which this wave has a row
    for (;;) {
      if (nx.rb + wave * g.rpw < nx.rhi) return;
      nx.job += gridDim.x;
      if (nx.job >= g.njobs) { nx.valid = false; return; }
      const Job jn = job_of(nx.job);
      nx.img = jn.img; nx.rhi = jn.rhi; nx.rb = jn.rlo;
    }
  };
  auto advance = [&]() {
    if (++nx.u < NU) return;
    nx.u = 0;
    if (++nx.j < g.nb) return;
    nx.j = 0;
    if (++nx.k < g.rpw && nx.rb + wave * g.rpw + nx.k < nx.rhi) return;
    nx.k = 0;
    nx.rb += phase_rows;
    settle();
  };
  // The B operand is fetched ONE unit ahead with two buffers (unit u of a block computes from xb[u & 1], NU is even), in two
  // halves: groups 0-3 of unit u + 1 at the top of unit u, groups 4-7 after MFMA group 3.  ALWAYS the same loads (past the last
  // unit: out of the descriptor's bound, no memory access): hipcc's s_waitcnt insertion takes the minimum over the arms of a
  // conditional fetch, and with "no loads" as one arm every MFMA group waits for the loads issued just before it.
  // (Fetching two units ahead — one load after every MFMA group into the register group it has just freed — measured 108
  // instead of 79 us: sixteen scattered loads per wave in flight fill the CU's vector-memory queue, a wave that cannot issue
  // its next load cannot issue its next MFMA either.)
  f32x4 xb[2][NJ];
  struct Src { __amdgpu_buffer_rsrc_t rsrc; int voff; };
  auto source = [&]() __attribute__((always_inline)) -> Src {          // of the unit the cursor points at
    const bool on = nx.valid && !(abl & 2);
    const int r = on ? nx.rb + wave * g.rpw + nx.k : 0;
    const float* const row = p.in + ((size_t)(on ? nx.img : 0) * p.h + r) * p.w * F;
    return Src{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, on ? (unsigned)(p.w * F * 4) : 0u, 0x00020000),
               ((32 * nx.j + pix) * F + half * (F / 2) + 32 * nx.u) * 4};
  };
  auto fetch1 = [&](const Src& sr, int jj) __attribute__((always_inline)) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sr.rsrc, sr.voff + 16 * jj, 0, 0));
  };
  {
    nx.job = blockIdx.x; nx.valid = nx.job < g.njobs;
    nx.k = nx.j = nx.u = 0; nx.img = 0; nx.rhi = 0; nx.rb = 0;
    if (nx.valid) {
      const Job j0 = job_of(nx.job);
      nx.img = j0.img; nx.rhi = j0.rhi; nx.rb = j0.rlo;
      settle();
    }
    const Src sr = source();                      // unit 0; the cursor stays on it until unit 0 fetches unit 1
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) xb[0][jj] = fetch1(sr, jj);
  }

  const int cout = p.cout_real;
  const int img_pix = p.h * p.w;
  // Second stage: a phase's outputs are numbered (row yy, channel co < CO, padded column); element tid + 512k of a thread is
  // decomposed once: off = its offset in the NCHW image from row yout on (out of every bound for padding), pq = its column in
  // the Q planes (co * wpad + x < 4096) | co << 12 | yy << 16
  int off[NPRE], pq[NPRE];
  {   // weights and bias to LDS: every load in flight before the first store, the decomposition meanwhile
    constexpr int WR = W_FLOATS / 4 / THREADS;            // 1 .. 8
    f32x4 wr[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) wr[i] = reinterpret_cast<const f32x4*>(p.wpk)[tid + i * THREADS];
    const float bv = tid < cout ? p.bias[tid] : 0.f;
    const float rcp_nb = 1.0f / (float)g.nb;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int b = (tid + k * THREADS) >> 5;
      const int t = (int)(((float)b + 0.5f) * rcp_nb);          // b / nb, exact for these sizes
      const int x = 32 * (b - t * g.nb) + (tid & 31);
      const int yy = t / CO, co = t - yy * CO;
      const bool real = x < p.w && co < cout;
      off[k] = real ? (co * img_pix + yy * p.w + x) * 4 : (int)0x80000000;
      pq[k] = real ? (co * g.wpad + x) | co << 12 | yy << 16 : 0;
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) reinterpret_cast<f32x4*>(w_s)[tid + i * THREADS] = wr[i];
    if (tid < 8) b_s[tid] = bv;
  }
  __syncthreads();
  stamp(1);
  const unsigned w_lane = (unsigned)(size_t)w_s + lane * 16;

  stamp(2);
  int stamp_i = 3;
  for (int job = blockIdx.x; job < g.njobs; job += gridDim.x) {
    const Job J = job_of(job);
    int yout = J.ys;
    for (int rb = J.rlo; rb < J.rhi; rb += phase_rows) {
      // ---- the rows this phase completes, and their skip values (in flight under the MFMAs) ----
      stamp(stamp_i);
      const int rc = min(rb + phase_rows, J.rhi);
      int ylim = rc == J.rhi ? J.ye : min(J.ye, rc - 1);
      if (ylim < yout) ylim = yout;
      const int rows = ylim - yout;
      const size_t img_off = (size_t)J.img * cout * img_pix + (size_t)yout * p.w;
      const unsigned img_bytes = (unsigned)(cout * img_pix - yout * p.w) * 4u;
      const float* const aux_img = p.aux + img_off;
      float* const out_img = p.out + img_off;
      const auto aux_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(aux_img), 0, img_bytes, 0x00020000);
      const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(out_img, 0, img_bytes, 0x00020000);
      const int lim = (abl & 1) ? 0 : rows * CO * g.wpad;          // elements of this phase
      float pre[NPRE];
#pragma unroll
      for (int k = 0; k < NPRE; ++k)
        pre[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                               aux_rsrc, tid + k * THREADS < lim ? off[k] : (int)0x80000000, 0, 0));

      stamp(stamp_i + 1);
      // ---- P for this wave's rows of the phase ----
      for (int k = 0; k < g.rpw; ++k) {
        const int r = rb + wave * g.rpw + k;
        if (r >= J.rhi) break;
        float* const q_row = q_s + (size_t)(r % g.ring) * (3 * CO) * g.wpad + (half * CL) * g.wpad + pix;
        float T[NH], P2h[NH], c0[NH];
#pragma unroll
        for (int i = 0; i < NH; ++i) T[i] = P2h[i] = c0[i] = 0.f;
        for (int j = 0; j < g.nb; ++j) {
          f32x16 acc[NRB];
#pragma unroll
          for (int b = 0; b < NRB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            if (nx.valid) advance();
            const Src sr = source();                  // the next unit of the stream
            f32x4 (&xc)[NJ] = xb[u & 1], (&xn)[NJ] = xb[(u + 1) & 1];
#pragma unroll
            for (int jj = 0; jj < NJ / 2; ++jj) xn[jj] = fetch1(sr, jj);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(abl & 4)) {
            const unsigned wb_u = w_lane + u * (NJ * 1024);
            f32x4 wa[2], wb[2];
#define OUTM_READ(J_, SLOT)                                                                                            \
  do {                                                                                                                  \
    if (NRB == 2)                                                                                                       \
      asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"                                     \
                   : "=&v"(wa[SLOT]), "=&v"(wb[SLOT]) : "v"(wb_u), "n"((J_) * 1024), "n"(NU * NJ * 1024 + (J_) * 1024));   \
    else                                                                                                                \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(wa[SLOT]) : "v"(wb_u), "n"((J_) * 1024));                    \
  } while (0)
#define OUTM_STEP(J_)                                                                                                   \
  do {                                                                                                                  \
    if (NRB == 2) {                                                                                                     \
      if ((J_) < NJ - 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(wa[(J_) & 1]), "+v"(wb[(J_) & 1]));                     \
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wa[(J_) & 1]), "+v"(wb[(J_) & 1]));                               \
    } else {                                                                                                            \
      if ((J_) < NJ - 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wa[(J_) & 1]));                                         \
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wa[(J_) & 1]));                                                   \
    }                                                                                                                   \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                                     \
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[(J_) & 1][e], xc[J_][e], acc[0], 0, 0, 0);                       \
      if (NRB == 2) acc[NRB - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[(J_) & 1][e], xc[J_][e], acc[NRB - 1], 0, 0, 0); \
    }                                                                                                                   \
    if ((J_) + 2 < NJ) OUTM_READ(((J_) + 2) & (NJ - 1), (J_) & 1);                                                            \
    if ((J_) == NJ / 2 - 1) {                                                                                           \
      _Pragma("unroll") for (int jj = NJ / 2; jj < NJ; ++jj) xn[jj] = fetch1(sr, jj);                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                                \
    }                                                                                                                   \
  } while (0)
            OUTM_READ(0, 0);
            OUTM_READ(1, 1);
            OUTM_STEP(0); OUTM_STEP(1); OUTM_STEP(2); OUTM_STEP(3); OUTM_STEP(4); OUTM_STEP(5); OUTM_STEP(6); OUTM_STEP(7);
#undef OUTM_STEP
#undef OUTM_READ
            } else {
#pragma unroll
              for (int jj = NJ / 2; jj < NJ; ++jj) xn[jj] = fetch1(sr, jj);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          // ---- the dx sum: this block's left part, the previous block's right part ----
          if (abl & 8) continue;
          float Tn[NH], P2n[NH], c0n[NH], nxt[NH], P0e[NH];
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int c = 0; c < CL; ++c) {
              const int hi = dy * CL + c;
              const int q0 = (dy * 3 + 0) * CL + c, q1 = (dy * 3 + 1) * CL + c, q2 = (dy * 3 + 2) * CL + c;
              const float P0 = acc[q0 >> 4][q0 & 15], P1 = acc[q1 >> 4][q1 & 15], P2 = acc[q2 >> 4][q2 & 15];
              const float sh0 = __builtin_bit_cast(
                  float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, P0), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
              Tn[hi] = (edge_l ? c0[hi] : sh0) + P1;
              P0e[hi] = P0;
              P2n[hi] = P2;
            }
#pragma unroll
          for (int i = 0; i < NH; ++i) c0n[i] = nxt[i] = 0.f;
          if (g.nb > 1) {                      // a row of one block has no neighbours to hand edge values to
#pragma unroll
            for (int i = 0; i < NH; ++i) {
              c0n[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(last_addr, __builtin_bit_cast(int, P0e[i])));
              nxt[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(first_addr, __builtin_bit_cast(int, P2n[i])));
            }
          }
          if (j > 0) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
              for (int c = 0; c < CL; ++c) {
                const int hi = dy * CL + c;
                const float sh2 = __builtin_bit_cast(
                    float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, P2h[hi]), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
                q_row[(dy * CO + c) * g.wpad + 32 * (j - 1)] = T[hi] + (edge_r ? nxt[hi] : sh2);
              }
          }
#pragma unroll
          for (int i = 0; i < NH; ++i) { T[i] = Tn[i]; P2h[i] = P2n[i]; c0[i] = c0n[i]; }
        }
        // the row's last block: nothing to its right
        if (!(abl & 8))
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int c = 0; c < CL; ++c) {
            const int hi = dy * CL + c;
            const float sh2 = __builtin_bit_cast(
                float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, P2h[hi]), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
            q_row[(dy * CO + c) * g.wpad + 32 * (g.nb - 1)] = T[hi] + (edge_r ? 0.f : sh2);
          }
      }
      stamp(stamp_i + 2);
      if (!(abl & 16)) __syncthreads();
      stamp(stamp_i + 3);

      // ---- the dy sum, bias, skip ----
      const int plane = 3 * CO * g.wpad;
      const int s0 = (yout + g.ring - 1) % g.ring;            // ring slot of row yout - 1
      auto q_sum = [&](int qcol, int yy) __attribute__((always_inline)) -> float {
        const int y = yout + yy;
        int sa = s0 + yy;                                     // slots of rows y - 1, y, y + 1 (yy < ring)
        sa = sa >= g.ring ? sa - g.ring : sa;
        int sb = sa + 1;
        sb = sb >= g.ring ? sb - g.ring : sb;
        int sc = sb + 1;
        sc = sc >= g.ring ? sc - g.ring : sc;
        const float* const qc = q_s + qcol;
        const float r0 = qc[sa * plane], a1 = qc[sb * plane + CO * g.wpad], r2 = qc[sc * plane + 2 * CO * g.wpad];
        const float a0 = y > 0 ? r0 : 0.f, a2 = y + 1 < p.h ? r2 : 0.f;
        return (a0 + a1) + a2;
      };
#pragma unroll
      for (int k = 0; k < NPRE; ++k) {
        const bool valid = tid + k * THREADS < lim;
        const int e = valid ? pq[k] : 0;
        const float v = (q_sum(e & 0xfff, e >> 16) + b_s[(e >> 12) & 7]) + pre[k];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), out_rsrc, valid ? off[k] : (int)0x80000000, 0, 0);
      }
      for (int idx = tid + NPRE * THREADS; idx < lim; idx += THREADS) {      // a phase of more than NPRE elements per thread
        const int b = idx >> 5;
        const int t = b / g.nb, x = 32 * (b - t * g.nb) + (tid & 31);
        const int yy = t / CO, co = t - yy * CO;
        if (x < p.w && co < cout) {
          const int o = co * img_pix + yy * p.w + x;
          out_img[o] = (q_sum(co * g.wpad + x, yy) + b_s[co]) + aux_img[o];
        }
      }
      yout = ylim;
      stamp(stamp_i + 4);
      if (!(abl & 16)) __syncthreads();
      stamp(stamp_i + 5);
      stamp_i += 6;
    }
  }
  stamp(31);
}

// ---- host side ----

static bool out_mfma_geometry(const ConvParams& p, int feat, int cus, OutMfmaGeom* g, size_t* lds_bytes) {
  if (p.cout_real < 1 || p.cout_real > 6 || (feat != 128 && feat != 256)) return false;
  if (p.n <= 0 || p.h <= 0 || p.w <= 0 || p.w > 4095 || (long long)p.w * feat * 4 > 0x7fffffffLL ||
      (long long)p.cout_real * p.h * p.w > 0x7fffffffLL) return false;
  const int cl = p.cout_real <= 2 ? 1 : 3, nrb = cl == 3 ? 2 : 1;
  g->nb = (p.w + 31) / 32;
  g->wpad = 32 * g->nb;
  g->rpw = g->wpad <= 32 ? 4 : g->wpad <= 64 ? 2 : 1;
  g->ring = 8 * g->rpw + 2;
  *lds_bytes = ((size_t)nrb * (feat / 64) * 2048 + 8 + (size_t)g->ring * 3 * (2 * cl) * g->wpad) * sizeof(float);
  if (*lds_bytes > outm::LDS_LIMIT) return false;
  // few images: cut them into strips of rows (each recomputes one row above and below) until every CU has two jobs
  const int phase_rows = 8 * g->rpw;
  const long long want_jobs = 2LL * cus;
  g->strip_rows = p.h;
  g->strips = 1;
  if (p.n < want_jobs && p.h > phase_rows) {
    const int want = (int)((want_jobs + p.n - 1) / p.n);
    int sr = (p.h + want - 1) / want;
    sr = (sr + 2 + phase_rows - 1) / phase_rows * phase_rows - 2;          // interior strips compute whole phases
    if (sr < p.h) { g->strip_rows = sr; g->strips = (p.h + sr - 1) / sr; }
  }
  const long long njobs = (long long)p.n * g->strips;
  if (njobs > 0x7fffffffLL) return false;
  g->njobs = (int)njobs;
  return true;
}

template <int F, int CL>
static hipError_t launch_out_mfma_one(const ConvParams& p, hipStream_t stream, bool* taken, int ablate) {
  auto kern = conv3x3_out_mfma_kernel<F, CL>;
  static KernelOnce once;
  int cus = 0;
  hipError_t e = once.prepare(reinterpret_cast<const void*>(kern), outm::LDS_LIMIT, &cus);
  if (e != hipSuccess) return e;
  OutMfmaGeom g;
  size_t lds = 0;
  if (!out_mfma_geometry(p, F, cus, &g, &lds)) { *taken = false; return hipSuccess; }
  *taken = true;
  g.ablate = ablate;
  const unsigned grid = (unsigned)(g.njobs < cus ? g.njobs : cus);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(outm::THREADS), lds, stream, p, g);
  return hipGetLastError();
}

// p.wpk = the weights packed by pack_out_mfma_weights_host.  *taken = false (and nothing launched) when the shape does not
// fit this kernel (row of Q too wide for LDS, more than 6 outputs): the caller then uses conv3x3_out.hip.
hipError_t launch_conv3x3_out_mfma(const ConvParams& p, int feat, hipStream_t stream, bool* taken, int ablate) {
  *taken = false;
  if (p.cout_real < 1 || p.cout_real > 6) return hipSuccess;
  const bool small = p.cout_real <= 2;
  if (feat == 128) return small ? launch_out_mfma_one<128, 1>(p, stream, taken, ablate) : launch_out_mfma_one<128, 3>(p, stream, taken, ablate);
  if (feat == 256) return small ? launch_out_mfma_one<256, 1>(p, stream, taken, ablate) : launch_out_mfma_one<256, 3>(p, stream, taken, ablate);
  return hipSuccess;
}

size_t out_mfma_weight_floats(int cin) { return (size_t)2 * (cin / 64) * 2048; }

// kernel HWIO (3,3,cin,cout<=6) -> [blk][u][jj][kk][m][e]: the A operand of MFMA step (u, jj < 8, e) for lane (m, kk) of row
// block blk = K[tap][kk*cin/2 + 32u + 4jj + e][co], where row m of block blk is accumulator register i = 4*(m>>3) + (m&3) of
// lane half h = (m>>2)&1, i.e. that half's value number q = 16*blk + i = (tap*CL + c), co = h*CL + c (CL = 1 for cout <= 2,
// else 3); zero where q >= 9*CL or co >= cout.  out_mfma_weight_floats(cin) floats (the second row block unused for CL = 1).
void pack_out_mfma_weights_host(const float* k, int cin, int cout, float* dst) {
  const int cl = cout <= 2 ? 1 : 3, nrb = cl == 3 ? 2 : 1, nu = cin / 64;
  const size_t total = out_mfma_weight_floats(cin);
  for (size_t i = 0; i < total; ++i) dst[i] = 0.f;
  if (cout > 6) return;
  for (int blk = 0; blk < nrb; ++blk)
    for (int u = 0; u < nu; ++u)
      for (int jj = 0; jj < 8; ++jj)
        for (int kk = 0; kk < 2; ++kk)
          for (int m = 0; m < 32; ++m) {
            const int h = (m >> 2) & 1, i = 4 * (m >> 3) + (m & 3), q = 16 * blk + i;
            if (q >= 9 * cl) continue;
            const int tap = q / cl, co = h * cl + (q - tap * cl);
            if (co >= cout) continue;
            for (int e = 0; e < 4; ++e) {
              const int c = kk * (cin / 2) + 32 * u + 4 * jj + e;
              dst[((((size_t)(blk * nu + u) * 8 + jj) * 2 + kk) * 32 + m) * 4 + e] = k[((size_t)tap * cin + c) * cout + co];
            }
          }
}

}  // namespace dsen2
