// Calibration, not product: what the matrix pipe of THIS board sustains under its power cap with nothing else going on.
//   hipcc --offload-arch=gfx950 -O3 -o build/mfma_peak tools/mfma_peak.hip && build/mfma_peak [seconds_per_variant [variant]]
// One workgroup of 512 threads per CU-slot (2 waves per SIMD, as the body convolutions run), every wave keeps the
// register tile of conv3x3_body16w (4 A fragments x 8 B fragments -> 32 accumulators of v_mfma_f32_16x16x32_bf16) and
// issues MFMAs back to back from registers: no LDS, no memory.  Operands are random bf16 (or zeros: the pipe's power
// depends on the data).  Prints TFLOP/s and the clock the kernel saw (s_memtime cycles per s_memrealtime 100 MHz tick).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int ORDER>
__global__ __launch_bounds__(512, 2) void bf16_loop(const uint4* __restrict__ src, float* __restrict__ sink, int iters,
                                                    unsigned long long* __restrict__ clk) {
    const int t = threadIdx.x;
    bf16x8 a[4], b[8];
    for (int i = 0; i < 4; ++i) { uint4 v = src[(t * 12 + i) & 4095]; a[i] = *reinterpret_cast<bf16x8*>(&v); }
    for (int i = 0; i < 8; ++i) { uint4 v = src[(t * 12 + 4 + i) & 4095]; b[i] = *reinterpret_cast<bf16x8*>(&v); }
    f32x4 acc[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            // 0: A fragment shared by 8 consecutive MFMAs (the product kernel's order); 1: B fragment shared by 4;
            // 2: both operands change with every MFMA
            const int i = ORDER == 0 ? n / 8 : ORDER == 1 ? n % 4 : n % 4;
            const int j = ORDER == 0 ? n % 8 : ORDER == 1 ? n / 4 : (n / 4 + n % 4 * 2 + n) % 8;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456f) sink[t] = s;
    if (blockIdx.x == 0 && t == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

// the 32x32x16 form of the same tile: 2 A fragments x 4 B fragments -> 8 accumulators of 16 registers
__global__ __launch_bounds__(512, 2) void bf16_loop_32(const uint4* __restrict__ src, float* __restrict__ sink, int iters,
                                                       unsigned long long* __restrict__ clk) {
    const int t = threadIdx.x;
    bf16x8 a[2], b[4];
    for (int i = 0; i < 2; ++i) { uint4 v = src[(t * 12 + i) & 4095]; a[i] = *reinterpret_cast<bf16x8*>(&v); }
    for (int i = 0; i < 4; ++i) { uint4 v = src[(t * 12 + 4 + i) & 4095]; b[i] = *reinterpret_cast<bf16x8*>(&v); }
    f32x16 acc[2][4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int k = 0; k < 16; ++k) s += acc[i][j][k];
    if (s == 123.456f) sink[t] = s;
    if (blockIdx.x == 0 && t == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

__global__ __launch_bounds__(512, 2) void f32_loop(const float* __restrict__ src, float* __restrict__ sink, int iters,
                                                   unsigned long long* __restrict__ clk) {
    const int t = threadIdx.x;
    float a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = src[(t * 6 + i) & 16383];
    for (int i = 0; i < 2; ++i) b[i] = src[(t * 6 + 4 + i) & 16383];
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) s += acc[i][j][k];
    if (s == 123.456f) sink[t] = s;
    if (blockIdx.x == 0 && t == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

// The same register tile fed from LDS: NREAD fragment reads (ds_read_b128, conflict-free, 1 KiB per wave each) per 32
// MFMAs.  conv3x3_body16w reads 12 per 32 (4 weight + 8 pixel fragments per step-quarter); 7 models a dx-major tap order
// that keeps 10 pixel-row fragments for three taps.
template <int NREAD>
__global__ __launch_bounds__(512, 2) void bf16_lds_loop(const uint4* __restrict__ src, float* __restrict__ sink, int iters,
                                                        unsigned long long* __restrict__ clk) {
    __shared__ uint4 lds[4096];                                   // 64 KiB
    const int t = threadIdx.x;
    for (int i = t; i < 4096; i += 512) lds[i] = src[i];
    __syncthreads();
    bf16x8 f[12];
    for (int i = 0; i < 12; ++i) { uint4 v = src[(t * 12 + i) & 4095]; f[i] = *reinterpret_cast<bf16x8*>(&v); }
    f32x4 acc[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lane = t & 63, wave = t >> 6;
    // fragments are rolled in place like the product kernel's: the read that refills a fragment for the next iteration
    // is issued right after the fragment's last use (A fragment i after block i, B fragment j inside block 3), so
    // every read has at least seven MFMAs of this wave (and the other wave's) to land.  NREAD = 12: all; 7: the four
    // A fragments + three B; 4: the A fragments only.
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        const int base = ((it * 5 + wave * 3) & 15) * 64 + lane;  // 16 windows of 64 slots (< 1024); + j * 256 < 4096
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[i], f[4 + j], acc[i][j], 0, 0, 0);
                if (i == 3 && 4 + j < NREAD) {
                    __builtin_amdgcn_sched_barrier(0);
                    uint4 v = lds[base + (4 + j) * 256];
                    f[4 + j] = *reinterpret_cast<bf16x8*>(&v);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (i < NREAD && i < 3) {
                __builtin_amdgcn_sched_barrier(0);
                uint4 v = lds[base + i * 256];
                f[i] = *reinterpret_cast<bf16x8*>(&v);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (3 < NREAD) {
            __builtin_amdgcn_sched_barrier(0);
            uint4 v = lds[base + 3 * 256];
            f[3] = *reinterpret_cast<bf16x8*>(&v);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456f) sink[t] = s;
    if (blockIdx.x == 0 && t == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

static uint16_t to_bf16(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x8000u) >> 16); }

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    const char* only = argc > 2 ? argv[2] : nullptr;          // run one variant (tools/power_probe.py --mfma)
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::vector<uint16_t> h(4096 * 8); std::vector<float> hf(16384);
    srand(1);
    auto rnd = []() { float s = 0; for (int i = 0; i < 12; ++i) s += rand() / (float)RAND_MAX; return s - 6.0f; };
    for (auto& v : h) v = to_bf16(rnd());
    for (auto& v : hf) v = rnd();
    void *d_rand, *d_zero, *d_f, *d_fz; float* sink; unsigned long long* clk;
    CHECK(hipMalloc(&d_rand, h.size() * 2)); CHECK(hipMalloc(&d_zero, h.size() * 2));
    CHECK(hipMalloc(&d_f, hf.size() * 4)); CHECK(hipMalloc(&d_fz, hf.size() * 4));
    CHECK(hipMalloc(&sink, 4096)); CHECK(hipMalloc(&clk, 16));
    CHECK(hipMemcpy(d_rand, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemset(d_zero, 0, h.size() * 2));
    CHECK(hipMemcpy(d_f, hf.data(), hf.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemset(d_fz, 0, hf.size() * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct V { const char* name; int kind; void* src; double flop_per_iter_wave; };
    const V vs[] = {
        {"bf16_16x16x32_random", 0, d_rand, 32.0 * 2 * 16 * 16 * 32},
        {"bf16_16x16x32_zeros", 0, d_zero, 32.0 * 2 * 16 * 16 * 32},
        {"bf16_16x16x32_random_b_outer", 101, d_rand, 32.0 * 2 * 16 * 16 * 32},
        {"bf16_16x16x32_random_diagonal", 102, d_rand, 32.0 * 2 * 16 * 16 * 32},
        {"bf16_32x32x16_random", 32, d_rand, 16.0 * 2 * 32 * 32 * 16},
        {"bf16_lds_reads_12_per_32", 12, d_rand, 32.0 * 2 * 16 * 16 * 32},
        {"bf16_lds_reads_7_per_32", 7, d_rand, 32.0 * 2 * 16 * 16 * 32},
        {"bf16_lds_reads_4_per_32", 4, d_rand, 32.0 * 2 * 16 * 16 * 32},
        {"f32_32x32x2_random", 1, d_f, 8.0 * 2 * 32 * 32 * 2},
        {"f32_32x32x2_zeros", 1, d_fz, 8.0 * 2 * 32 * 32 * 2},
    };
    const int grid = cus * 1, iters = 20000;          // ~ms-long launches
    for (const V& v : vs) {
        if (only && strcmp(only, v.name) != 0) continue;
        double total_ms = 0; long launches = 0; unsigned long long hc[2] = {0, 0};
        // warm-up, then launches of the same kernel back to back for `seconds`
        for (int rep = 0; rep < 2; ++rep) {
            total_ms = 0; launches = 0;
            const double budget = rep == 0 ? 0.3 : seconds;
            while (total_ms < budget * 1e3) {
                CHECK(hipEventRecord(e0, 0));
                for (int k = 0; k < 20; ++k) {
                    if (v.kind == 32) bf16_loop_32<<<grid, 512>>>((const uint4*)v.src, sink, iters, clk);
                    else if (v.kind == 12) bf16_lds_loop<12><<<grid, 512>>>((const uint4*)v.src, sink, iters, clk);
                    else if (v.kind == 7) bf16_lds_loop<7><<<grid, 512>>>((const uint4*)v.src, sink, iters, clk);
                    else if (v.kind == 4) bf16_lds_loop<4><<<grid, 512>>>((const uint4*)v.src, sink, iters, clk);
                    else if (v.kind == 101) bf16_loop<1><<<grid, 512>>>((const uint4*)v.src, sink, iters, clk);
                    else if (v.kind == 102) bf16_loop<2><<<grid, 512>>>((const uint4*)v.src, sink, iters, clk);
                    else if (v.kind == 0) bf16_loop<0><<<grid, 512>>>((const uint4*)v.src, sink, iters, clk);
                    else f32_loop<<<grid, 512>>>((const float*)v.src, sink, iters, clk);
                }
                CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); total_ms += ms; launches += 20;
            }
        }
        CHECK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
        const double flop = v.flop_per_iter_wave * iters * 8.0 * grid * launches;
        printf("{\"variant\": \"%s\", \"cus\": %d, \"tflops\": %.1f, \"ms_per_launch\": %.4f, \"kernel_clock_ghz\": %.3f, \"seconds\": %.1f}\n",
               v.name, cus, flop / (total_ms * 1e-3) / 1e12, total_ms / launches, hc[1] ? (double)hc[0] / hc[1] * 0.1 : 0.0, total_ms * 1e-3);
        fflush(stdout);
    }
    return 0;
}
