// Which fp16 matrix-instruction shape should the f16x3 convolutions use?  Both v_mfma_f32_32x32x16_f16 and
// v_mfma_f32_16x16x32_f16 do 1024 multiply-adds per cycle and SIMD, but the chip is power-limited in these loops and may
// hold a different clock on each shape (MI355X_MICROARCH.md, "DVFS give-back" item 7: 1.12-1.15 x for bare bf16 loops).
// This probe runs the K loop of conv_pipe_kernel<3,2> in both shapes on random data:
//   per wave a 96-pixel x 64-channel fp32 accumulator tile, operands as two fp16 terms, three cross products;
//   A (pixels x channels) read from an LDS halo tile with ds_read_b128 at the tap's offset, B (packed weights) read from
//   global memory in fragment order (L1 / L2 hits, all four waves of a block read the same fragments);
//   operands of the next unit in flight while the matrix instructions of the current one issue.
// Same output tile, same bytes from LDS and from memory, same number of matrix cycles; only the instruction shape differs.
// Build: hipcc --offload-arch=gfx950 -O3 -o shape_probe shape_probe.hip      Run: ./shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

constexpr int TH = 16, TW = 24, HW = TW + 2, HH = TH + 2;   // 384 output pixels per block = 4 waves x 96
constexpr int HALO = HH * HW;                                // 468 staged pixels
constexpr int LP4 = 5;                                       // LDS pixel pitch in 16-byte units: [2 terms][16 ch] fp16 + 16 B pad
constexpr int PLANE4 = HALO * LP4;                           // one 16-channel plane, 16-byte units
constexpr int CIN = 64;                                      // K = 9 taps x 64 channels = 576

// ---- 32x32x16: unit = (tap, 16-channel plane); 3 x 2 accumulator tiles of 32 x 32 -------------------------------------
// ORDER: the issue order of the MT x NT x 3 products of a unit (per-accumulator order unchanged, so results are the same):
//   0 = product-major, then row tile, then column tile (A repeats NT times, B alternates)      1 = accumulator-major: the three products of
//   an accumulator back to back (a dependent chain, as conv_wgrad_wm_kernel issues them)      2 = product-major, column tile, row tile (B
//   repeats MT times)      3 = row tile, product, column tile (one A fragment through all its uses before the next)
template <int MT, int NT, int WPS, int ORDER = 0>
__global__ __launch_bounds__(256, WPS) void loop32(const uint4* __restrict__ seed, const uint4* __restrict__ wpk, float* __restrict__ out,
                                              int iters) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    for (int i = tid; i < PLANE4 * 2; i += 256) lds[i] = seed[i];
    __syncthreads();
    int lbase[MT];
    for (int i = 0; i < MT; ++i) {
        const int q = ((wave * MT + i) * 32 + r) % (TH * TW), ly = q / TW, lx = q - ly * TW;
        lbase[i] = (ly * HW + lx) * LP4 + h;
    }
    f32x16 acc[MT][NT];
    for (int i = 0; i < MT; ++i)
        for (int j = 0; j < NT; ++j)
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    constexpr int UNITS = 9 * (CIN / 16);
    uint4 a[2][2][MT], b[2][2][NT];     // [buffer][term][tile]
    auto load = [&](int buf, int u) {
        const int tap = u % 9, pl = u / 9;
        const int off = (pl & 1) * PLANE4 + ((tap / 3) * HW + tap % 3) * LP4;   // two resident planes, as the two-slot tile of the real kernel
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < MT; ++i) a[buf][s][i] = lds[lbase[i] + off + s * 2];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[buf][s][j] = wpk[((u * NT + j) * 2 + s) * 64 + lane];
        }
    };
    for (int it = 0; it < iters; ++it) {
        load(0, 0);
#pragma unroll 2
        for (int u = 0; u < UNITS; ++u) {
            const int cur = u & 1;
            if (u + 1 < UNITS) load(cur ^ 1, u + 1);
            auto mm = [&](int sa, int sb, int i, int j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[cur][sa][i]), __builtin_bit_cast(f16x8, b[cur][sb][j]),
                                                                  acc[i][j], 0, 0, 0);
                if constexpr (ORDER >= 0) __builtin_amdgcn_sched_barrier(0);      // pin the issue order (the scheduler rotates accumulators by itself)
            };
            constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};
            if constexpr (ORDER == 0) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) mm(PA[p], PB[p], i, j);
            } else if constexpr (ORDER == 1) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int p = 0; p < 3; ++p) mm(PA[p], PB[p], i, j);
            } else if constexpr (ORDER == 2) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int i = 0; i < MT; ++i) mm(PA[p], PB[p], i, j);
            } else {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
#pragma unroll
                        for (int j = 0; j < NT; ++j) mm(PA[p], PB[p], i, j);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < MT; ++i)
        for (int j = 0; j < NT; ++j)
            for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + tid] = s;
}

// ---- 16x16x32: unit = (tap, 32 channels = two planes, half of the wave's rows); 6 x 4 accumulator tiles of 16 x 16 ------
template <int MT, int NT, int WPS>      // MT, NT in units of 32: 2 MT x 2 NT tiles of 16 x 16
__global__ __launch_bounds__(256, WPS) void loop16(const uint4* __restrict__ seed, const uint4* __restrict__ wpk, float* __restrict__ out,
                                              int iters) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;       // row of the 16-row tile; 8-channel group of the 32-channel K step
    for (int i = tid; i < PLANE4 * 2; i += 256) lds[i] = seed[i];
    __syncthreads();
    int lbase[2 * MT];
    for (int i = 0; i < 2 * MT; ++i) {
        const int q = ((wave * 2 * MT + i) * 16 + r) % (TH * TW), ly = q / TW, lx = q - ly * TW;
        lbase[i] = (g >> 1) * PLANE4 + (ly * HW + lx) * LP4 + (g & 1);
    }
    f32x4 acc[2 * MT][2 * NT];
    for (int i = 0; i < 2 * MT; ++i)
        for (int j = 0; j < 2 * NT; ++j)
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    constexpr int STEPS = 9 * (CIN / 32);         // K steps of 32 channels
    uint4 a[2][2][MT], b[2][2][2 * NT];                 // A: [buffer][term][row tile of the half]; B: [buffer][term][column tile]
    auto load_a = [&](int buf, int st, int half) {
        const int tap = st % 9;
        const int off = ((tap / 3) * HW + tap % 3) * LP4;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i) a[buf][s][i] = lds[lbase[half * MT + i] + off + s * 2];
    };
    auto load_b = [&](int buf, int st) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 2 * NT; ++j) b[buf][s][j] = wpk[((st * 2 * NT + j) * 2 + s) * 64 + lane];
    };
    for (int it = 0; it < iters; ++it) {
        load_a(0, 0, 0);
        load_b(0, 0);
#pragma unroll 2
        for (int st = 0; st < STEPS; ++st) {
            const int bb = st & 1;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                // next unit's operands
                if (half == 0) load_a(1, st, 1);
                else if (st + 1 < STEPS) {
                    load_a(0, st + 1, 0);
                    load_b(bb ^ 1, st + 1);
                }
#pragma unroll
                for (int sa = 0; sa < 2; ++sa)
#pragma unroll
                    for (int sb = 0; sb < 2 - sa; ++sb)
#pragma unroll
                        for (int i = 0; i < MT; ++i)
#pragma unroll
                            for (int j = 0; j < 2 * NT; ++j)
                                acc[half * MT + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                                    __builtin_bit_cast(f16x8, a[half][sa][i]), __builtin_bit_cast(f16x8, b[bb][sb][j]), acc[half * MT + i][j], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2 * MT; ++i)
        for (int j = 0; j < 2 * NT; ++j)
            for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + tid] = s;
}

static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

template <int MT, int NT, int WPS>
static void run_config(const uint4* dseed, const uint4* dw, float* dout, size_t lds_bytes, const char* what) {
    const int nblocks = 256 * 8, iters = 40;
    auto k32 = loop32<MT, NT, WPS>;
    auto k16 = loop16<MT, NT, WPS>;
    CHECK(hipFuncSetAttribute((const void*)k32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    CHECK(hipFuncSetAttribute((const void*)k16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const double flop = (double)nblocks * 4 * MT * 32 * NT * 32 * 9 * CIN * 2 * 3 * iters;   // issued fp16 flops (3 products)
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%s: wave tile %d x %d, 9 taps x %d channels, f16x3 (3 products), %d blocks x %d iterations, LDS %zu B per block\n", what, MT * 32,
           NT * 32, CIN, nblocks, iters, lds_bytes);
    for (int round = 0; round < 2; ++round) {
        for (int shape = 0; shape < 2; ++shape) {
            // ~2 s of back-to-back launches first: the clock the chip settles at is what matters
            float ms = 0.f;
            int reps = 0;
            for (int phase = 0; phase < 2; ++phase) {
                const int n = phase == 0 ? 1000 : 100;
                CHECK(hipEventRecord(e0));
                for (int k = 0; k < n; ++k) {
                    if (shape == 0) hipLaunchKernelGGL(k32, dim3(nblocks), dim3(256), lds_bytes, 0, dseed, dw, dout, iters);
                    else hipLaunchKernelGGL(k16, dim3(nblocks), dim3(256), lds_bytes, 0, dseed, dw, dout, iters);
                }
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                reps = n;
            }
            printf("  round %d  %s  %.3f ms per launch  %.1f TFLOP/s of fp16 issue (%.1f of fp32 work)\n", round,
                   shape == 0 ? "32x32x16" : "16x16x32", ms / reps, flop / (ms / reps * 1e-3) * 1e-12, flop / 3 / (ms / reps * 1e-3) * 1e-12);
            fflush(stdout);
        }
    }
    CHECK(hipGetLastError());
}

template <int MT, int NT, int WPS, int ORDER>
static void run_order(const uint4* dseed, const uint4* dw, float* dout, size_t lds_bytes) {
    const int nblocks = 256 * 8, iters = 40;
    auto k = loop32<MT, NT, WPS, ORDER>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const double flop = (double)nblocks * 4 * MT * 32 * NT * 32 * 9 * CIN * 2 * 3 * iters;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float ms = 0.f;
    int reps = 0;
    for (int phase = 0; phase < 2; ++phase) {
        const int n = phase == 0 ? 600 : 100;
        CHECK(hipEventRecord(e0));
        for (int q = 0; q < n; ++q) hipLaunchKernelGGL(k, dim3(nblocks), dim3(256), lds_bytes, 0, dseed, dw, dout, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        reps = n;
    }
    printf("  order %d  %.3f ms per launch  %.1f TFLOP/s of fp16 issue\n", ORDER, ms / reps, flop / (ms / reps * 1e-3) * 1e-12);
    fflush(stdout);
}
template <int MT, int NT, int WPS>
static void run_orders(const uint4* dseed, const uint4* dw, float* dout, size_t lds_bytes, const char* what) {
    printf("%s\n", what);
    for (int round = 0; round < 2; ++round) {
        run_order<MT, NT, WPS, 0>(dseed, dw, dout, lds_bytes);
        run_order<MT, NT, WPS, 1>(dseed, dw, dout, lds_bytes);
        run_order<MT, NT, WPS, 2>(dseed, dw, dout, lds_bytes);
        run_order<MT, NT, WPS, 3>(dseed, dw, dout, lds_bytes);
    }
}

int main() {
    const size_t lds2 = (size_t)PLANE4 * 2 * 16;       // 75 KB: two blocks per CU, as conv_pipe_kernel
    // random fp16 operands: high terms uniform in (-1, 1), low terms 2^-11 of that (what split2h produces)
    std::vector<_Float16> hseed(lds2 / 2), hw((size_t)9 * (CIN / 16) * 2 * 2 * 64 * 8);
    srand(7);
    for (size_t i = 0; i < hseed.size(); ++i) hseed[i] = (_Float16)(((i % 40) / 16 == 1) ? frand() * 4.8e-4f : frand());
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (_Float16)(((i / 512) & 1) ? frand() * 4.8e-4f : frand());
    uint4 *dseed, *dw;
    float* dout;
    CHECK(hipMalloc(&dseed, lds2));
    CHECK(hipMalloc(&dw, hw.size() * 2));
    CHECK(hipMalloc(&dout, (size_t)256 * 8 * 256 * 4));
    CHECK(hipMemcpy(dseed, hseed.data(), lds2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    // (a) the register tile of conv_pipe_kernel<3,2>, ONE wave per SIMD for both shapes (the 16x16x32 form needs > 256 registers
    //     with both operand sets double-buffered; the LDS request keeps the 32x32x16 form at one block per CU too)
    run_config<3, 2, 1>(dseed, dw, dout, 100 * 1024, "(a) one wave per SIMD");
    // (b) a 64 x 64 register tile, TWO waves per SIMD for both shapes
    run_config<2, 2, 2>(dseed, dw, dout, lds2, "(b) two waves per SIMD");
    // (c) issue order of the products of a unit, 32x32x16, both tiles (round 4: does operand reuse between consecutive matrix instructions
    //     change what the power envelope leaves?)
    run_orders<3, 2, 1>(dseed, dw, dout, 100 * 1024, "(c) one wave per SIMD, 96 x 64");
    run_orders<2, 2, 2>(dseed, dw, dout, lds2, "(c) two waves per SIMD, 64 x 64");
    return 0;
}
